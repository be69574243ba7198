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

/* ---- recorder taps on the reference's own decode (used by tests/golden/make_golden.py::gen_hevc_file) ----
 * The leaf calls of decode_intra_block (hevc.c:4665-4805) are `static`; they are tapped at their CALL SITES by
 * function-like macros that dispatch on __LINE__ (gcc expands it to the line the invocation starts on): at the line of
 * the definition the macro yields the function's own name (a macro's own name is not expanded again), at the line of
 * the call in decode_intra_block it yields the recorder, which notes the arguments and forwards to the real function.
 * No reference source is changed or copied; the line numbers below are build-recipe knowledge of the pinned tree. */
#include "hevc.h"
struct picture;
struct cu;
struct sps;
struct slice_segment_header;
struct hevc_param_set;
#define REF_CAT_(a, b) a##b
#define REF_CAT(a, b) REF_CAT_(a, b)
static void rec_intra_sample_prediction(struct slice_segment_header *slice, struct hevc_param_set *hps, struct cu *cu, int xTbCmp,
                                        int yTbCmp, int predModeIntra, int nTbS, int cIdx, int16_t *predSamples, struct picture *p);
static int rec_scale_and_transform(struct cu *cu, int transform_skip_flag, struct hevc_param_set *hps, struct slice_segment_header *slice,
                                   int xTbY, int yTbY, int cIdx, int nTbS, int16_t *r, struct picture *p);
static void rec_scale_transform_coefficients(struct sps *sps, struct cu *cu, struct slice_segment_header *slice, struct picture *p,
                                             int transform_skip_flag, int xTbY, int yTbY, int nTbS, int cIdx, int qP, int16_t *d);
static void rec_reference_sample_substitution(struct sps *sps, int16_t *left, int16_t *top, int nTbS, int cIdx, int unavaible,
                                              int8_t *unavaibleL, int8_t *unavaibleT);
static void rec_rdpcm(int mDir, int nTbs, int16_t *r);
static void rec_construct_pic(struct sps *sps, int xCurr, int yCurr, int nCurrSw, int nCurrSh, int cIdx, int16_t *predSamples,
                              int16_t *resSamples, int16_t *dst, int stride);
static void rec_yuv420_16(uint8_t *ptr, int pitch, int16_t *y, int16_t *u, int16_t *v, int y_stride, int uv_stride, int ctbrows, int ctbcols, int ctbsize);
#define intra_sample_prediction(...) REF_CAT(ISP_AT_, __LINE__)(__VA_ARGS__)
#define ISP_AT_4542(...) intra_sample_prediction(__VA_ARGS__)
#define ISP_AT_4731(...) rec_intra_sample_prediction(__VA_ARGS__)
#define scale_and_transform(...) REF_CAT(SAT_AT_, __LINE__)(__VA_ARGS__)
#define SAT_AT_4172(...) scale_and_transform(__VA_ARGS__)
#define SAT_AT_4738(...) rec_scale_and_transform(__VA_ARGS__)
#define scale_transform_coefficients(...) REF_CAT(STC_AT_, __LINE__)(__VA_ARGS__)
#define STC_AT_3743(...) scale_transform_coefficients(__VA_ARGS__)
#define STC_AT_4224(...) rec_scale_transform_coefficients(__VA_ARGS__)
#define reference_sample_substitution(...) REF_CAT(RSS_AT_, __LINE__)(__VA_ARGS__)
#define RSS_AT_4277(...) reference_sample_substitution(__VA_ARGS__)
#define RSS_AT_4623(...) rec_reference_sample_substitution(__VA_ARGS__)
#define residual_modification_transform_bypass(...) REF_CAT(RMB_AT_, __LINE__)(__VA_ARGS__)
#define RMB_AT_3960(...) residual_modification_transform_bypass(__VA_ARGS__)
#define RMB_AT_4746(...) rec_rdpcm(__VA_ARGS__)
#define construct_pic_pior_to_filtering(...) REF_CAT(CPP_AT_, __LINE__)(__VA_ARGS__)
#define CPP_AT_4256(...) construct_pic_pior_to_filtering(__VA_ARGS__)
#define CPP_AT_4791(...) rec_construct_pic(__VA_ARGS__)
#define YUV420_to_BGRA32_16bit(...) REF_CAT(Y16_AT_, __LINE__)(__VA_ARGS__)
#define Y16_AT_7261(...) rec_yuv420_16(__VA_ARGS__)
#include "hevc.c" /* the reference's coding/hevc.c */
#undef intra_sample_prediction
#undef scale_and_transform
#undef scale_transform_coefficients
#undef reference_sample_substitution
#undef residual_modification_transform_bypass
#undef construct_pic_pior_to_filtering
#undef YUV420_to_BGRA32_16bit


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


/* ------------------------------------------------------------------------------------------------------------
 * The recorder behind the taps above.  While ref_hevc_record_begin() ... ref_hevc_record_end() brackets a decode, every
 * leaf transform unit the reference reconstructs leaves one record, in decode order: what ffhip_hevc_residual_batch and
 * ffhip_hevc_intra_recon take as input (TU geometry, mode, flags, availability, quantised levels, qP) and what the
 * reference computed from it (the residual block after scale_and_transform, the planes and the BGRA picture). */
struct rec_tu {
    int32_t x, y, log2, cidx, mode, flags; /* flags: ffhip FFHIP_TU_* bits */
    int32_t qp, rflags;                    /* residual stage: qP, 1 = DST (intra luma 4x4), 2 = transform skip, 4 = bypass, 8 = rotate */
    int32_t level_off;                     /* element offset into the level / residual arrays, -1 = no residual */
    int32_t pad;
    uint64_t avail_top, avail_left;
};
static struct {
    int on, open;
    struct rec_tu cur, *tus;
    long n_tus, cap_tus;
    int16_t *levels, *resid;
    long n_lv, cap_lv;
    int16_t *planes;  /* copy of the picture planes handed to the colour conversion */
    long plane_elems;
    int y_stride, uv_stride, ctbrows, ctbcols, ctbsize, pitch;
    struct sps *sps;
    struct cu *cu;
} g_rec;

static void rec_intra_sample_prediction(struct slice_segment_header *slice, struct hevc_param_set *hps, struct cu *cu, int xTbCmp,
                                        int yTbCmp, int predModeIntra, int nTbS, int cIdx, int16_t *predSamples, struct picture *p)
{
    if (g_rec.on) {
        struct pps *pps = hps->pps[slice->slice_pic_parameter_set_id];
        struct sps *sps = hps->sps[pps->pps_seq_parameter_set_id];
        memset(&g_rec.cur, 0, sizeof g_rec.cur);
        g_rec.cur.x = xTbCmp; g_rec.cur.y = yTbCmp; g_rec.cur.log2 = log2floor(nTbS); g_rec.cur.cidx = cIdx; g_rec.cur.mode = predModeIntra;
        g_rec.cur.level_off = -1;
        g_rec.cur.avail_top = nTbS == 32 ? ~0ull : (1ull << (2 * nTbS)) - 1; /* everything available unless the substitution is invoked */
        g_rec.cur.avail_left = g_rec.cur.avail_top;
        int fl = 1; /* corner */
        if (sps->sps_range_ext.intra_smoothing_disabled_flag == 0 && (cIdx == 0 || sps->ChromaArrayType == 3)) fl |= 4;
        if (sps->strong_intra_smoothing_enabled_flag) fl |= 8;
        if (sps->sps_scc_ext.intra_boundary_filtering_disabled_flag == 1) fl |= 16 | 32;
        else if (sps->sps_range_ext.implicit_rdpcm_enabled_flag == 1 && cu->cu_transquant_bypass_flag == 1) fl |= 16;
        g_rec.cur.flags = fl;
        g_rec.open = 1;
        g_rec.sps = sps;
        g_rec.cu = cu;
    }
    intra_sample_prediction(slice, hps, cu, xTbCmp, yTbCmp, predModeIntra, nTbS, cIdx, predSamples, p);
}

/* the reference's function as it is, for drivers that keep their own record open (ref_hevc_isp_picture) */
static void rec_intra_sample_prediction_bare(struct slice_segment_header *slice, struct hevc_param_set *hps, struct cu *cu, int xTbCmp,
                                             int yTbCmp, int predModeIntra, int nTbS, int cIdx, int16_t *predSamples, struct picture *p)
{
    intra_sample_prediction(slice, hps, cu, xTbCmp, yTbCmp, predModeIntra, nTbS, cIdx, predSamples, p);
}

static void rec_reference_sample_substitution(struct sps *sps, int16_t *left, int16_t *top, int nTbS, int cIdx, int unavaible,
                                              int8_t *unavaibleL, int8_t *unavaibleT)
{
    if (g_rec.on && g_rec.open) {
        uint64_t at = 0, al = 0;
        for (int k = 0; k < 2 * nTbS; k++) {
            if (!unavaibleT[k]) at |= 1ull << k;
            if (!unavaibleL[k]) al |= 1ull << k;
        }
        g_rec.cur.avail_top = at;
        g_rec.cur.avail_left = al;
        if (unavaibleT[-1]) g_rec.cur.flags &= ~1;
    }
    reference_sample_substitution(sps, left, top, nTbS, cIdx, unavaible, unavaibleL, unavaibleT);
}

static void rec_scale_transform_coefficients(struct sps *sps, struct cu *cu, struct slice_segment_header *slice, struct picture *p,
                                             int transform_skip_flag, int xTbY, int yTbY, int nTbS, int cIdx, int qP, int16_t *d)
{
    if (g_rec.on && g_rec.open) g_rec.cur.qp = qP;
    scale_transform_coefficients(sps, cu, slice, p, transform_skip_flag, xTbY, yTbY, nTbS, cIdx, qP, d);
}

static int rec_scale_and_transform(struct cu *cu, int transform_skip_flag, struct hevc_param_set *hps, struct slice_segment_header *slice,
                                   int xTbY, int yTbY, int cIdx, int nTbS, int16_t *r, struct picture *p)
{
    long off = -1;
    if (g_rec.on && g_rec.open) {
        struct pps *pps = hps->pps[slice->slice_pic_parameter_set_id];
        struct sps *sps = hps->sps[pps->pps_seq_parameter_set_id];
        const int nn = nTbS * nTbS;
        if (g_rec.n_lv + nn > g_rec.cap_lv) {
            g_rec.cap_lv = 2 * (g_rec.cap_lv + nn) + 4096;
            g_rec.levels = realloc(g_rec.levels, (size_t)g_rec.cap_lv * 2);
            g_rec.resid = realloc(g_rec.resid, (size_t)g_rec.cap_lv * 2);
        }
        off = g_rec.n_lv;
        struct trans_tree *tt = &cu->tt;
        for (int y = 0; y < nTbS; y++)
            for (int x = 0; x < nTbS; x++) g_rec.levels[off + x + y * nTbS] = tt->TransCoeffLevel[cIdx][xTbY + x - tt->xT0][yTbY + y - tt->yT0];
        g_rec.n_lv += nn;
        g_rec.cur.level_off = (int32_t)off;
        g_rec.cur.flags |= 2;
        const int intra = get_CuPredMode(sps, p, xTbY, yTbY) == MODE_INTRA;
        g_rec.cur.rflags = ((cIdx == 0 && nTbS == 4 && intra) ? 1 : 0) | (transform_skip_flag ? 2 : 0) | (cu->cu_transquant_bypass_flag ? 4 : 0) |
                           ((sps->sps_range_ext.transform_skip_rotation_enabled_flag == 1 && nTbS == 4 && intra) ? 8 : 0);
    }
    const int rc = scale_and_transform(cu, transform_skip_flag, hps, slice, xTbY, yTbY, cIdx, nTbS, r, p);
    if (off >= 0) memcpy(g_rec.resid + off, r, (size_t)nTbS * nTbS * 2);
    return rc;
}

static void rec_rdpcm(int mDir, int nTbs, int16_t *r)
{
    if (g_rec.on && g_rec.open) g_rec.cur.flags |= 64;
    residual_modification_transform_bypass(mDir, nTbs, r);
}

static void rec_construct_pic(struct sps *sps, int xCurr, int yCurr, int nCurrSw, int nCurrSh, int cIdx, int16_t *predSamples,
                              int16_t *resSamples, int16_t *dst, int stride)
{
    construct_pic_pior_to_filtering(sps, xCurr, yCurr, nCurrSw, nCurrSh, cIdx, predSamples, resSamples, dst, stride);
    if (g_rec.on && g_rec.open) {
        if (g_rec.n_tus == g_rec.cap_tus) {
            g_rec.cap_tus = 2 * g_rec.cap_tus + 1024;
            g_rec.tus = realloc(g_rec.tus, (size_t)g_rec.cap_tus * sizeof *g_rec.tus);
        }
        if (xCurr != g_rec.cur.x || yCurr != g_rec.cur.y || cIdx != g_rec.cur.cidx || nCurrSw != (1 << g_rec.cur.log2)) g_rec.cur.pad = 1; /* mismatch marker */
        g_rec.tus[g_rec.n_tus++] = g_rec.cur;
        g_rec.open = 0;
    }
}

static void rec_yuv420_16(uint8_t *ptr, int pitch, int16_t *y, int16_t *u, int16_t *v, int y_stride, int uv_stride, int ctbrows, int ctbcols, int ctbsize)
{
    if (g_rec.on) { /* planes: Y at 0, U at size, V at size * 3/2 of one allocation of 2 * size samples (hevc.c:7225-7230) */
        g_rec.plane_elems = (long)(u - y) * 2;
        g_rec.planes = realloc(g_rec.planes, (size_t)g_rec.plane_elems * 2);
        memcpy(g_rec.planes, y, (size_t)g_rec.plane_elems * 2);
        g_rec.y_stride = y_stride; g_rec.uv_stride = uv_stride; g_rec.ctbrows = ctbrows; g_rec.ctbcols = ctbcols; g_rec.ctbsize = ctbsize; g_rec.pitch = pitch;
        (void)v;
    }
    YUV420_to_BGRA32_16bit(ptr, pitch, y, u, v, y_stride, uv_stride, ctbrows, ctbcols, ctbsize);
}

void ref_hevc_record_begin(void)
{
    g_rec.on = 1; g_rec.open = 0; g_rec.n_tus = 0; g_rec.n_lv = 0;
}
/* info[8] = n_tus, n_level_elements, plane elements (Y + U + V region), y_stride, uv_stride, ctbrows, ctbcols, ctbsize */
void ref_hevc_record_end(long *info)
{
    g_rec.on = 0;
    info[0] = g_rec.n_tus; info[1] = g_rec.n_lv; info[2] = g_rec.plane_elems; info[3] = g_rec.y_stride; info[4] = g_rec.uv_stride;
    info[5] = g_rec.ctbrows; info[6] = g_rec.ctbcols; info[7] = g_rec.ctbsize;
}
void ref_hevc_record_fetch(void *tus, int16_t *levels, int16_t *resid, int16_t *planes)
{
    memcpy(tus, g_rec.tus, (size_t)g_rec.n_tus * sizeof *g_rec.tus);
    memcpy(levels, g_rec.levels, (size_t)g_rec.n_lv * 2);
    memcpy(resid, g_rec.resid, (size_t)g_rec.n_lv * 2);
    if (g_rec.plane_elems) memcpy(planes, g_rec.planes, (size_t)g_rec.plane_elems * 2);
}
/* sps facts the test needs to size its buffers: out[6] = width, height, BitDepthY, BitDepthC, ChromaArrayType, CtbLog2SizeY */
void ref_hevc_sps_info(void *hps_, int *out)
{
    struct hevc_param_set *hps = hps_;
    struct sps *sps = hps->sps[0];
    out[0] = sps->pic_width_in_luma_samples; out[1] = sps->pic_height_in_luma_samples; out[2] = sps->BitDepthY; out[3] = sps->BitDepthC;
    out[4] = sps->ChromaArrayType; out[5] = sps->CtbLog2SizeY;
}
void *ref_hevc_param_set_new(void) { return calloc(1, sizeof(struct hevc_param_set)); }

/* ---- config 5's CPU baseline (bench.py `extra.c5.cpu_baseline`): ONE picture through the three post-entropy stages in C,
 * one call per picture, every arithmetic step the reference's own code: per TU in decode order
 * scale_transform_coefficients + transform_scaled_coeffients (ref_hevc_scale / ref_hevc_transform above), then the
 * neighbour processing, prediction and construct_pic_pior_to_filtering of ref_hevc_intra_tu; at the end
 * YUV420_to_BGRA32_16bit over the planes, as parse_slice_segment_layer does (hevc.c:7260-7277).
 * tus: ffhip_hevc_tu records (32 bytes each); levels and residual share ONE layout (element offset res_offset). */
struct ref_chain_tu { uint16_t x, y; uint8_t log2_size, cidx, pred_mode, flags; uint32_t res_offset; int32_t res_scale; uint64_t avail_top, avail_left; };
void ref_hevc_chain_picture(const struct ref_chain_tu *tus, long n_tus, const int16_t *levels, int16_t *residual, int qP,
                            int bitdepth, int16_t *py, int16_t *pu, int16_t *pv, int width, int height, int ctbsize,
                            uint8_t *bgra, int pitch)
{
    int16_t d[32 * 32];
    for (long i = 0; i < n_tus; i++) {
        const struct ref_chain_tu *t = tus + i;
        const int n = 1 << t->log2_size;
        if (t->flags & 2) {
            ref_hevc_scale(levels + t->res_offset, d, n, qP, bitdepth, 0, NULL, t->cidx);
            ref_hevc_transform(d, residual + t->res_offset, n, t->cidx == 0 && n == 4, bitdepth, 0);
        }
        ref_hevc_intra_tu(t->x, t->y, t->log2_size, t->cidx, t->pred_mode, t->flags, t->avail_top, t->avail_left,
                          residual + t->res_offset, t->cidx == 0 ? py : (t->cidx == 1 ? pu : pv), t->cidx == 0 ? width : width / 2,
                          bitdepth, bitdepth, t->res_scale);
    }
    YUV420_to_BGRA32_16bit(bgra, pitch, py, pu, pv, width, width / 2, height / ctbsize, width / ctbsize, ctbsize);
}

/* ---- the reference's static intra_sample_prediction ITSELF (hevc.c:4542-4662) over a caller-prepared TU list -------------
 * hps: a parameter set the reference's own parse_nalu filled from VPS / SPS / PPS NAL units (so MinTbAddrZs, CtbAddrRsToTs,
 * TileId and the picture geometry are the reference's); the picture is allocated as parse_slice_segment_layer does
 * (hevc.c:7221-7240).  Per TU in list order: CuPredMode of the TU's coding block is set (what parse_coding_unit has done by
 * the time decode_intra_block runs; it decides availability under constrained_intra_pred_flag), intra_sample_prediction
 * gathers the neighbours with ITS OWN process_zscan_order_block_availablity, substitutes, filters and predicts; the
 * availability it derived is read back through the call-site tap on reference_sample_substitution; the residual is added by
 * construct_pic_pior_to_filtering.  Nothing of the neighbour gathering is restated here.
 * masks_out[2 i], [2 i + 1] = avail_top, avail_left of TU i, corner_out[i] = its corner availability.
 * Returns the number of int16 elements of the picture (Y | U | V as at hevc.c:7225-7230) copied to pixel_out, or -1. */
long ref_hevc_isp_picture(void *hps_, const struct ref_chain_tu *tus, long n_tus, const int16_t *residual, int16_t *pixel_out,
                          long pixel_cap, uint64_t *masks_out, uint8_t *corner_out, int *geom_out)
{
    struct hevc_param_set *hps = hps_;
    struct slice_segment_header *slice = calloc(1, sizeof *slice);
    slice->slice_pic_parameter_set_id = 0;
    struct pps *pps = hps->pps[0];
    struct sps *sps = hps->sps[pps->pps_seq_parameter_set_id];
    /* what the slice header parser does once it knows the parameter sets (hevc.c:2701-2702): the derived picture geometry
     * and the z-scan / tile tables process_zscan_order_block_availablity reads */
    calc_sps_params(sps);
    calc_pps_params(sps, pps);
    const int width = sps->pic_width_in_luma_samples, height = ((sps->pic_height_in_luma_samples + 3) >> 2) << 2;
    const int y_stride = ((width + 3) >> 2) << 2, uv_stride = y_stride >> 1;
    struct picture p;
    memset(&p, 0, sizeof p);
    p.size = height * y_stride;
    p.pixel = calloc((size_t)height * y_stride * 2, sizeof(int16_t));
    p.y_stride = y_stride;
    p.uv_stride = uv_stride;
    p.info = calloc((size_t)sps->PicWidthInMinCbsY * sps->PicHeightInMinCbsY, sizeof(struct cu_info));
    struct cu *cu = calloc(1, sizeof *cu);
    long rc = -1;
    if (!p.pixel || !p.info || !cu || (long)p.size * 2 > pixel_cap) goto out;
    for (long i = 0; i < n_tus; i++) {
        const struct ref_chain_tu *t = tus + i;
        const int nTbS = 1 << t->log2_size, cIdx = t->cidx;
        if (cIdx == 0) {
            const int lg = t->log2_size > sps->MinCbLog2SizeY ? t->log2_size : sps->MinCbLog2SizeY;
            set_CuPredMode(sps, &p, (t->x >> lg) << lg, (t->y >> lg) << lg, lg, MODE_INTRA);
        }
        int16_t predSamples[64 * 64];
        int16_t resSamples[32 * 32] = {0};
        memset(&g_rec.cur, 0, sizeof g_rec.cur);
        g_rec.cur.avail_top = nTbS == 32 ? ~0ull : (1ull << (2 * nTbS)) - 1; /* everything available unless the substitution is invoked */
        g_rec.cur.avail_left = g_rec.cur.avail_top;
        g_rec.cur.flags = 1;
        g_rec.on = 1; g_rec.open = 1;
        rec_intra_sample_prediction_bare(slice, hps, cu, t->x, t->y, t->pred_mode, nTbS, cIdx, predSamples, &p);
        g_rec.on = 0; g_rec.open = 0;
        masks_out[2 * i] = g_rec.cur.avail_top;
        masks_out[2 * i + 1] = g_rec.cur.avail_left;
        corner_out[i] = (uint8_t)(g_rec.cur.flags & 1);
        if (t->flags & 2) {
            memcpy(resSamples, residual + t->res_offset, (size_t)nTbS * nTbS * sizeof(int16_t));
            if (t->flags & 64) residual_modification_transform_bypass(t->pred_mode / 26, nTbS, resSamples);
        }
        int16_t *dst = cIdx == 0 ? p.pixel : (cIdx == 1 ? p.pixel + p.size : p.pixel + p.size * 3 / 2);
        construct_pic_pior_to_filtering(sps, t->x, t->y, nTbS, nTbS, cIdx, predSamples, resSamples, dst, cIdx == 0 ? y_stride : uv_stride);
    }
    memcpy(pixel_out, p.pixel, (size_t)p.size * 2 * sizeof(int16_t));
    geom_out[0] = width; geom_out[1] = height; geom_out[2] = y_stride; geom_out[3] = uv_stride; geom_out[4] = p.size;
    geom_out[5] = pps->constrained_intra_pred_flag; geom_out[6] = sps->strong_intra_smoothing_enabled_flag;
    geom_out[7] = sps->sps_range_ext.intra_smoothing_disabled_flag;
    rc = (long)p.size * 2;
out:
    free(p.pixel); free(p.info); free(cu); free(slice);
    return rc;
}
