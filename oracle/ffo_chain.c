/*
 * oracle/ffo_chain.c -- whole-frame / whole-picture loops over the restated stages, one call per frame.
 * TEST INFRASTRUCTURE ONLY (see oracle/ffo.h): what bench.py times as the "port" CPU baseline of configs 4 and 5 when
 * the compiled reference (oracle/_ref) is not there, and what the parity flags of its batch sweeps compare with.
 *
 * Follows (reference file:line, /root/reference):
 *   ffo_vp8_chain_frame      format/webp.c:1833-1868   the frame loop of vp8_decode: residual -> predict -> loop filter -> colour
 *   ffo_hevc_chain_picture   coding/hevc.c:4730-4791, 7260-7277   decode_intra_block's leaf calls per TU, colour at the end
 */
#include "ffo.h"

#include <stddef.h>

void ffo_vp8_chain_frame(int mbcols, int mbrows, const int16_t *levels, const uint8_t *info, const uint16_t *quant,
                         const uint8_t *modes, int filter_type, const uint8_t *filters, int16_t *residual, uint8_t *yp,
                         uint8_t *up, uint8_t *vp, uint8_t *bgra, int pitch)
{
    const long n_mb = (long)mbcols * mbrows;
    for (long i = 0; i < n_mb; i++)
        ffo_vp8_residual_mb(levels + i * 400, info + i * 32, info[i * 32 + 25], quant + 8 * (info[i * 32 + 26] & 3), residual + i * 384);
    ffo_vp8_recon_frame(mbcols, mbrows, modes, residual, NULL, yp, up, vp);
    if (filter_type) ffo_vp8_loopfilter_frame(mbcols, mbrows, filter_type, modes, filters, yp, up, vp);
    ffo_yuv420_to_bgra32(bgra, pitch, yp, up, vp, 16 * mbcols, 8 * mbcols, mbrows, mbcols);
}

void ffo_hevc_chain_picture(const ffo_hevc_tu *tus, long n_tus, const int16_t *levels, int16_t *residual, int qP, int bitdepth,
                            int16_t *py, int16_t *pu, int16_t *pv, int width, int height, int ctbsize, uint8_t *bgra, int pitch)
{
    for (long i = 0; i < n_tus; i++) {
        const ffo_hevc_tu *t = tus + i;
        const int n = 1 << t->log2_size;
        if (t->flags & FFO_TU_RESIDUAL)
            ffo_hevc_residual_tu(levels + t->res_offset, residual + t->res_offset, n, qP, (t->cidx == 0 && n == 4) ? 1 : 0, bitdepth, 0, NULL);
        ffo_hevc_intra_tu(t, residual, t->cidx == 0 ? py : (t->cidx == 1 ? pu : pv), t->cidx == 0 ? width : width / 2, bitdepth, bitdepth);
    }
    ffo_yuv420_to_bgra32_16bit(bgra, pitch, py, pu, pv, width, width / 2, height / ctbsize, width / ctbsize, ctbsize);
}
