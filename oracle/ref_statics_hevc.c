/*
 * oracle/ref_statics_hevc.c -- part of the oracle/_ref build recipe.
 * TEST INFRASTRUCTURE ONLY; compiles only where /root/reference exists.
 *
 * Compiles the reference's own coding/hevc.c inside this translation unit and
 * exports wrappers around its `static` hot-path functions
 *   scale_transform_coefficients  (hevc.c:3743-3816)
 *   transform_scaled_coeffients   (hevc.c:3888-3956)
 * building just enough decoder state (one intra CU at the picture origin) for
 * them to run, and one around scale_and_transform itself (hevc.c:4172-4251) with the qP
 * derivation in front of it.  Block layout at this interface is row-major d[x + y*nTbS].
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

/* Neighbour processing + prediction + reconstruction of one TU through the reference's own
 * reference_sample_substitution (hevc.c:4277-4351), filtering_neighbouring_samples
 * (hevc.c:4355-4426), hevc_intra_planar/DC/angular (format/predict.c:651-792),
 * residual_modification_transform_bypass (hevc.c:3960-3977) and
 * construct_pic_pior_to_filtering (hevc.c:4252-4274).  The gathering loop mirrors
 * intra_sample_prediction (hevc.c:4570-4608) with the availability decisions supplied by the
 * caller instead of process_zscan_order_block_availablity.
 * flags: 1 corner, 2 residual, 4 filter, 8 strong, 16 no_bf, 32 no_dc_bf, 64 rdpcm,
 * 128 cross-component prediction through residual_modification_transform_cross_prediction
 * (hevc.c:3979-3988) with the argument aliasing of its call site (hevc.c:4753-4755). */
void ref_hevc_intra_tu(int x0, int y0, int log2n, int cIdx, int predModeIntra, int flags, uint64_t avail_top,
                       uint64_t avail_left, int16_t *res_in, int16_t *dst, int stride, int bitdepth_y, int bitdepth_c,
                       int res_scale)
{
    struct sps *sps = calloc(1, sizeof *sps);
    const int nTbS = 1 << log2n;
    sps->BitDepthY = bitdepth_y;
    sps->BitDepthC = bitdepth_c;
    sps->strong_intra_smoothing_enabled_flag = (flags & 8) != 0;
    int unavaible = 0;
    int8_t unavaibleL[64] = {0}, unavaibleA[65] = {0};
    int8_t *unavaibleT = unavaibleA + 1;
    int16_t left_default[64] = {0};
    int16_t top_default[65] = {0};
    int16_t *top = top_default + 1;
    int16_t *left = left_default;
    for (int x = -1; x < nTbS * 2; x++) {
        int ok = x < 0 ? (flags & 1) : (int)((avail_top >> x) & 1);
        if (!ok) { unavaible++; unavaibleT[x] = 1; }
        else top[x] = dst[x0 + x + (y0 - 1) * stride];
    }
    for (int y = 0; y < nTbS * 2; y++) {
        if (!((avail_left >> y) & 1)) { unavaible++; unavaibleL[y] = 1; }
        else left[y] = dst[x0 - 1 + (y0 + y) * stride];
    }
    if (unavaible > 0) reference_sample_substitution(sps, left, top, nTbS, cIdx, unavaible, unavaibleL, unavaibleT);
    if (flags & 4) filtering_neighbouring_samples(sps, predModeIntra, cIdx, nTbS, left, top);
    int16_t predSamples[64 * 64];
    int16_t resSamples[32 * 32] = {0};
    if (predModeIntra == INTRA_PLANAR)
        hevc_intra_planar((uint16_t *)predSamples, (uint16_t *)left, (uint16_t *)top, nTbS, nTbS);
    else if (predModeIntra == INTRA_DC)
        hevc_intra_DC((uint16_t *)predSamples, (uint16_t *)left, (uint16_t *)top, nTbS, nTbS, cIdx, (flags & 32) != 0);
    else
        hevc_intra_angular((uint16_t *)predSamples, (uint16_t *)left, (uint16_t *)top, nTbS, nTbS, cIdx,
                           predModeIntra, (flags & 16) != 0, sps->BitDepthY);
    if (flags & 2) {
        memcpy(resSamples, res_in, nTbS * nTbS * sizeof(int16_t));
        if (flags & 64) residual_modification_transform_bypass(predModeIntra / 26, nTbS, resSamples);
        if (flags & 128) {
            struct cu *cu = calloc(1, sizeof *cu);
            cu->ccp[0][0].ResScaleVal[cIdx] = (uint32_t)res_scale;
            residual_modification_transform_cross_prediction(sps, cu, 0, 0, nTbS, cIdx, resSamples, resSamples);
            free(cu);
        }
    }
    construct_pic_pior_to_filtering(sps, x0, y0, nTbS, nTbS, cIdx, predSamples, resSamples, dst, stride);
    free(sps);
}

/* The reference's scale_and_transform (hevc.c:4172-4251) itself, with every branch it has: transquant bypass
 * (:4209-4222), transform skip with `<< tsShift` (:4229-4236), the 180-degree rotation of 4x4 intra blocks
 * (rotateCoeffs, :4203-4207), scaling lists dropped for transform-skipped blocks larger than 4x4 (:3786-3787), the
 * DST entry for intra luma 4x4 (:3907-3921) and the qP derivation of 8.6.1 (:3998-4168) in front of it.
 * Decoder state: one intra coding unit, one coding tree block, one slice, one tile, the TU at the picture origin;
 * qp is the SliceQpY + QpBdOffset the test wants for luma; the function returns the qP the reference derived for
 * cIdx from it (luma: qp; chroma: through its Table 8-10 / min(qPi, 51) mapping, plus QpBdOffsetC).
 * level, r: row-major [x + y*nTbS]; scaling_factor row-major or NULL (scaling_list_enabled_flag = 0). */
int ref_hevc_scale_and_transform(const int16_t *level, int16_t *r, int nTbS, int cIdx, int qp, int bitdepth, int epp,
                                 int bypass, int transform_skip, int rotation_enabled, int chroma_array_type,
                                 const uint8_t *scaling_factor)
{
    struct hevc_param_set *hps = calloc(1, sizeof *hps);
    struct pps *pps = calloc(1, sizeof *pps);
    struct sps *sps = calloc(1, sizeof *sps);
    struct slice_segment_header *slice = calloc(1, sizeof *slice);
    struct picture *p = calloc(1, sizeof *p);
    struct cu_info *info = calloc(1, sizeof *info);
    struct ctu *ctu = calloc(1, sizeof *ctu), *ctus[1] = {ctu};
    struct cu *cu = calloc(1, sizeof *cu);
    int zs_col[16] = {0}, *zs[16];
    for (int i = 0; i < 16; i++) zs[i] = zs_col;
    hps->pps[0] = pps;
    hps->sps[0] = sps;
    sps->BitDepthY = sps->BitDepthC = bitdepth;
    sps->QpBdOffsetY = sps->QpBdOffsetC = 6 * (bitdepth - 8);
    sps->sps_range_ext.extended_precision_processing_flag = epp;
    sps->sps_range_ext.transform_skip_rotation_enabled_flag = rotation_enabled;
    sps->scaling_list_enabled_flag = scaling_factor != NULL;
    sps->ChromaArrayType = chroma_array_type;
    sps->MinCbLog2SizeY = 6;
    sps->CtbLog2SizeY = 6;
    sps->CtbSizeY = 64;
    sps->MinTbLog2SizeY = 2;
    sps->PicWidthInCtbsY = 1;
    sps->PicWidthInMinCbsY = 1;
    sps->pic_width_in_luma_samples = sps->pic_height_in_luma_samples = 64;
    pps->init_qp_minus26 = qp - 6 * (bitdepth - 8) - 26; /* SliceQpY; slice_qp_delta and CuQpDeltaVal stay 0 */
    pps->MinTbAddrZs = zs;
    slice->Log2MinCuQpDeltaSize = 6;
    info->CuPredMode = MODE_INTRA;
    p->info = info;
    p->ctus = ctus;
    cu->cu_transquant_bypass_flag = bypass;
    cu->log2CbSize = 6;
    const int sizeid = log2floor(nTbS) - 2;
    for (int y = 0; y < nTbS; y++)
        for (int x = 0; x < nTbS; x++) {
            cu->tt.TransCoeffLevel[cIdx][x][y] = level[x + y * nTbS];
            if (scaling_factor) slice->ScalingFactor[sizeid][cIdx][x][y] = scaling_factor[x + y * nTbS];
        }
    const struct quant_pixel q = quatization_parameters(0, 0, hps, slice, cu, p);
    const int qP = cIdx == 0 ? clip3(0, 51 + sps->QpBdOffsetY, q.q_y) : (cIdx == 1 ? q.q_cb : q.q_cr);
    scale_and_transform(cu, transform_skip, hps, slice, 0, 0, cIdx, nTbS, r, p);
    free(cu); free(ctu); free(info); free(p); free(slice); free(sps); free(pps); free(hps);
    return qP;
}
