/*
 * oracle/ref_statics_hevc.c -- part of the oracle/_ref build recipe.
 * TEST INFRASTRUCTURE ONLY; compiles only where /root/reference exists.
 *
 * Compiles the reference's own coding/hevc.c inside this translation unit and
 * exports wrappers around its `static` hot-path functions
 *   scale_transform_coefficients  (hevc.c:3743-3816)
 *   transform_scaled_coeffients   (hevc.c:3888-3956)
 * building just enough decoder state (one intra CU at the picture origin) for
 * them to run.  Block layout at this interface is row-major d[x + y*nTbS].
 */
#include "hevc.c" /* the reference's coding/hevc.c */

struct ref_hevc_ctx {
    struct sps sps;
    struct picture pic;
    struct cu_info info;
    struct slice_segment_header slice;
    struct cu cu;
};

static struct ref_hevc_ctx *ctx_new(int bitdepth, int epp, int pred_mode)
{
    struct ref_hevc_ctx *c = calloc(1, sizeof *c);
    c->sps.BitDepthY = c->sps.BitDepthC = bitdepth;
    c->sps.sps_range_ext.extended_precision_processing_flag = epp;
    c->sps.MinCbLog2SizeY = 6;
    c->sps.PicWidthInMinCbsY = 1;
    c->info.CuPredMode = pred_mode;
    c->pic.info = &c->info;
    return c;
}

/* level, d: row-major [x + y*nTbS]; scaling_factor row-major or NULL (flat 16) */
void ref_hevc_scale(const int16_t *level, int16_t *d, int nTbS, int qP, int bitdepth, int epp,
                    const uint8_t *scaling_factor, int cIdx)
{
    struct ref_hevc_ctx *c = ctx_new(bitdepth, epp, MODE_INTRA);
    int sizeid = log2floor(nTbS) - 2;
    c->sps.scaling_list_enabled_flag = scaling_factor != NULL;
    for (int y = 0; y < nTbS; y++)
        for (int x = 0; x < nTbS; x++) {
            c->cu.tt.TransCoeffLevel[cIdx][x][y] = level[x + y * nTbS];
            if (scaling_factor) c->slice.ScalingFactor[sizeid][cIdx][x][y] = scaling_factor[x + y * nTbS];
        }
    scale_transform_coefficients(&c->sps, &c->cu, &c->slice, &c->pic, 0, 0, 0, nTbS, cIdx, qP, d);
    free(c);
}

/* luma_intra_4x4 != 0 selects the cIdx == 0 / MODE_INTRA / nTbS == 4 entry (DST);
 * with no accelerator registered that is idct_4x4_hevc (hevc.c:3917). */
void ref_hevc_transform(int16_t *d, int16_t *r, int nTbS, int luma_intra_4x4, int bitdepth, int epp)
{
    struct ref_hevc_ctx *c = ctx_new(bitdepth, epp, MODE_INTRA);
    transform_scaled_coeffients(&c->sps, &c->pic, 0, 0, nTbS, luma_intra_4x4 ? 0 : 1, d, r);
    free(c);
}
