/*
 * ffhip_entropy.c -- host-side JPEG front end feeding the batched reconstruction
 *  and the BMP sink behind it (row f2).  Plain C11, no HIP.
 *
 * Stands where these pieces of the reference stand (paths in the ffpic tree):
 *   marker loop, SOF/DQT/DHT/DRI/SOS parsing   format/jpg.c:78-105, 640-655, 771-855
 *   read_compressed_scan (FF00 unstuffing, RSTn) format/jpg.c:588-637
 *   decode_data_unit (baseline branch)         format/jpg.c:255-415
 *   huffman_decode_symbol                      coding/huffman.c:92-222
 *   restart-interval bookkeeping               format/jpg.c:562-573
 *   BMP writer (54-byte header, top-down 32 bit) display/bmpwriter.c:19-81
 *
 * It writes what the reconstruction stage reads: quantised coefficients, natural
 * (de-zigzagged) order, int16, blocks in MCU order per component, and natural-order
 * uint16 quant tables -- directly into caller-provided (pinned or plain) host buffers.
 * Baseline / extended-sequential Huffman, 8-bit, interleaved scans, chroma 1x1 (what the
 * reference's colour converter supports); anything else returns FFHIP_EINVAL so the caller
 * can keep its C path.  Decoding is written from ITU-T T.81, not from the reference's code:
 * on well-formed streams both produce the same coefficients (tests pin this through the
 * reference's whole-file decode); the reference's end-of-scan overrun (utils/bitstream.c:117)
 * is not reproduced.
 */
#include "ffpic_hip.h"
#include "ffhip_entropy_internal.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const uint8_t k_zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                                     12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                     35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                                     58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static int huff_build(struct huff *h, const uint8_t counts[16], const uint8_t *vals, int nvals)
{
    memset(h, 0, sizeof *h);
    int code = 0, k = 0;
    memcpy(h->vals, vals, (size_t)nvals);
    for (int len = 1; len <= 16; len++) {
        h->valptr[len] = k;
        h->mincode[len] = code;
        for (int i = 0; i < counts[len - 1]; i++, k++, code++) {
            if (k >= nvals) return -1;
            if (code >= (1 << len)) return -1; /* over-subscribed DHT (Kraft sum > 1): the code does not fit its length */
            if (len <= LOOK) {
                int first = code << (LOOK - len), n = 1 << (LOOK - len);
                for (int j = 0; j < n; j++) h->look[first + j] = (uint16_t)((len << 8) | vals[k]);
            }
        }
        h->maxcode[len] = counts[len - 1] ? code - 1 : -1;
        code <<= 1;
    }
    h->maxcode[17] = 0x7fffffff;
    h->present = 1;
    for (int i = 0; i < (1 << LOOK); i++) {
        const unsigned e = h->look[i];
        const int len = (int)(e >> 8), run = (int)((e >> 4) & 15), mag = (int)(e & 15);
        if (!e || !mag || len + mag > LOOK) continue;
        int k = ((i << len) & ((1 << LOOK) - 1)) >> (LOOK - mag);
        if (k < (1 << (mag - 1))) k -= (1 << mag) - 1; /* EXTEND of T.81 F.2.2.1 */
        if (k >= -128 && k <= 127) h->fast[i] = (int16_t)((k * 256) + (run * 16) + len + mag);
    }
    return 0;
}

struct bits {
    const uint8_t *p, *end;
    uint64_t acc;
    int n;      /* valid bits in acc */
    int marker; /* a marker was hit: feed zeros */
    int dry;    /* zero bytes fed behind the end of the data (look-ahead only, in a well-formed stream) */
};

static inline void bits_fill(struct bits *b)
{
    if (!b->marker && b->end - b->p >= 8 && b->n <= 56) {
        /* eight bytes at once when none of them is 0xFF (no stuffing, no marker): the common case */
        uint64_t w;
        memcpy(&w, b->p, 8);
        w = __builtin_bswap64(w);
        const uint64_t x = ~w; /* a 0xFF byte of w is a zero byte of x */
        if (!((x - 0x0101010101010101ULL) & ~x & 0x8080808080808080ULL)) {
            const int nb = (64 - b->n) >> 3;
            b->acc = nb == 8 ? w : (b->acc << (8 * nb)) | (w >> (64 - 8 * nb));
            b->p += nb;
            b->n += 8 * nb;
            return;
        }
    }
    while (b->n <= 56) {
        unsigned c = 0;
        int real = 0;
        if (!b->marker && b->p < b->end) {
            c = *b->p;
            if (c == 0xFF) {
                if (b->p + 1 < b->end && b->p[1] == 0) { b->p += 2; real = 1; } /* stuffed zero (jpg.c:588-637) */
                else { b->marker = 1; c = 0; }
            } else { b->p++; real = 1; }
        }
        b->dry += !real;
        b->acc = (b->acc << 8) | c;
        b->n += 8;
    }
}
static inline int bits_get(struct bits *b, int k)
{
    if (k == 0) return 0;
    if (b->n < k) bits_fill(b);
    b->n -= k;
    return (int)((b->acc >> b->n) & ((1u << k) - 1));
}
static inline int huff_decode(struct bits *b, const struct huff *h)
{
    if (b->n < 16) bits_fill(b);
    unsigned peek = (unsigned)((b->acc >> (b->n - LOOK)) & ((1u << LOOK) - 1));
    unsigned e = h->look[peek];
    if (e) { b->n -= (int)(e >> 8); return (int)(e & 0xff); }
    int code = (int)peek, len = LOOK;
    while (code > h->maxcode[len]) {
        if (++len > 16) return -1; /* no code of any length matches: corrupt data (b->n >= 16, so no shift below is negative) */
        code = (int)((b->acc >> (b->n - len)) & ((1u << len) - 1));
    }
    b->n -= len;
    return h->vals[(h->valptr[len] + code - h->mincode[len]) & 255]; /* & 255: a malformed DHT must not index outside the table */
}
static inline int extend(int v, int t) { return (t && v < (1 << (t - 1))) ? v - (1 << t) + 1 : v; }

static int parse_headers(const uint8_t *f, size_t len, struct jpeg_hdr *j)
{
    memset(j, 0, sizeof *j);
    for (int t = 0; t < 4; t++)
        for (int i = 0; i < 64; i++) j->quant[t][i] = 1;
    if (len < 4 || f[0] != 0xFF || f[1] != 0xD8) return FFHIP_EINVAL;
    size_t p = 2;
    int have_sof = 0;
    while (p + 4 <= len) {
        if (f[p] != 0xFF) return FFHIP_EINVAL;
        while (p < len && f[p] == 0xFF) p++; /* fill bytes */
        if (p >= len) return FFHIP_EINVAL;
        const int m = f[p++];
        if (m == 0xD9) return FFHIP_EINVAL; /* EOI before SOS */
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (p + 2 > len) return FFHIP_EINVAL;
        const size_t L = ((size_t)f[p] << 8) | f[p + 1];
        if (L < 2 || p + L > len) return FFHIP_EINVAL;
        const uint8_t *s = f + p + 2;
        const size_t sl = L - 2;
        p += L;
        if (m == 0xDB) { /* DQT: stored de-zigzagged like read_dqt (jpg.c:78-105) */
            size_t i = 0;
            while (i < sl) {
                const int prec = s[i] >> 4, id = s[i] & 15;
                i++;
                if (id > 3 || i + (size_t)64 * (prec + 1) > sl) return FFHIP_EINVAL;
                for (int k = 0; k < 64; k++, i += prec + 1)
                    j->quant[id][k_zigzag[k]] = prec ? (uint16_t)((s[i] << 8) | s[i + 1]) : s[i];
            }
        } else if (m == 0xC4) { /* DHT */
            size_t i = 0;
            while (i + 17 <= sl) {
                const int tc = s[i] >> 4, th = s[i] & 15;
                int n = 0;
                for (int k = 0; k < 16; k++) n += s[i + 1 + k];
                if (tc > 1 || th > 3 || n > 256 || i + 17 + (size_t)n > sl) return FFHIP_EINVAL;
                if (huff_build(tc ? &j->ac[th] : &j->dc[th], s + i + 1, s + i + 17, n)) return FFHIP_EINVAL;
                i += 17 + (size_t)n;
            }
        } else if (m == 0xC0 || m == 0xC1) { /* SOF0 / SOF1 */
            if (sl < 6 || s[0] != 8) return FFHIP_EINVAL;
            j->height = (s[1] << 8) | s[2];
            j->width = (s[3] << 8) | s[4];
            j->ncomp = s[5];
            if ((j->ncomp != 1 && j->ncomp != 3) || sl < (size_t)(6 + 3 * j->ncomp)) return FFHIP_EINVAL;
            if (j->width == 0 || j->height == 0) return FFHIP_EINVAL; /* height 0 = "see DNL": not this path */
            for (int c = 0; c < j->ncomp; c++) {
                j->cid[c] = s[6 + 3 * c];
                j->h[c] = s[7 + 3 * c] >> 4;
                j->v[c] = s[7 + 3 * c] & 15;
                j->tq[c] = s[8 + 3 * c];
                if (j->tq[c] > 3) return FFHIP_EINVAL;
            }
            have_sof = 1;
        } else if (m == 0xC2 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
            return FFHIP_EINVAL; /* progressive / lossless / arithmetic: not this path */
        } else if (m == 0xDD) {
            if (sl < 2) return FFHIP_EINVAL;
            j->restart = (s[0] << 8) | s[1];
        } else if (m == 0xDA) { /* SOS */
            if (!have_sof || sl < 1 || s[0] != j->ncomp || sl < (size_t)(4 + 2 * j->ncomp)) return FFHIP_EINVAL;
            for (int k = 0; k < j->ncomp; k++) {
                int c;
                for (c = 0; c < j->ncomp && j->cid[c] != s[1 + 2 * k]; c++) {}
                if (c == j->ncomp) return FFHIP_EINVAL;
                j->td[c] = s[2 + 2 * k] >> 4;
                j->ta[c] = s[2 + 2 * k] & 15;
                if (j->td[c] > 3 || j->ta[c] > 3 || !j->dc[j->td[c]].present || !j->ac[j->ta[c]].present) return FFHIP_EINVAL;
            }
            const uint8_t *t = s + 1 + 2 * j->ncomp;
            if (t[0] != 0 || t[1] != 63 || t[2] != 0) return FFHIP_EINVAL;
            j->scan = f + p;
            j->scan_len = len - p;
            break;
        }
    }
    if (!j->scan) return FFHIP_EINVAL;
    if (j->ncomp == 1) { j->h[0] = j->v[0] = 1; } /* single-component scans are never interleaved */
    if (j->h[0] < 1 || j->v[0] < 1 || j->h[0] * j->v[0] > 4) return FFHIP_EINVAL; /* jpg.c:501: Y[3][64*4] */
    for (int c = 1; c < j->ncomp; c++)
        if (j->h[c] != 1 || j->v[c] != 1) return FFHIP_EINVAL; /* colorspace.c:149-150: chroma is one block per MCU */
    return FFHIP_OK;
}

/* for the other translation units of the library (ffhip_huff_gpu.hip) */
int ffhip_jpeg_parse(const uint8_t *file, size_t len, struct jpeg_hdr *j) { return parse_headers(file, len, j); }
/* the DRI value of a file this front end accepts (0 = none), -1 if it does not parse */
int ffhip_jpeg_probe_restart(const uint8_t *file, size_t len)
{
    struct jpeg_hdr *j = malloc(sizeof *j);
    if (!j) return -1;
    const int r = parse_headers(file, len, j) ? -1 : j->restart;
    free(j);
    return r;
}

int ffhip_jpeg_probe(const uint8_t *file, size_t len, ffhip_jpeg_geom *geom, int *width, int *height)
{
    struct jpeg_hdr *j = malloc(sizeof *j);
    if (!j) return FFHIP_ENOMEM;
    int rc = file && geom ? parse_headers(file, len, j) : FFHIP_EINVAL;
    if (rc == FFHIP_OK) {
        geom->ncomp = j->ncomp;
        geom->h = j->h[0];
        geom->v = j->v[0];
        geom->mcu_cols = (j->width + 8 * j->h[0] - 1) / (8 * j->h[0]);
        geom->mcu_rows = (j->height + 8 * j->v[0] - 1) / (8 * j->v[0]);
        for (int c = 0; c < 3; c++) geom->qt_id[c] = c < j->ncomp ? j->tq[c] : 0;
        if (width) *width = j->width;
        if (height) *height = j->height;
    }
    free(j);
    return rc;
}

/* `count` MCUs starting at `mcu`, from a bit reader positioned at the start of their entropy-coded
 * segment, DC predictors zero (start of scan or just behind an RSTn): the loop of read_compressed_scan /
 * decode_data_unit (jpg.c:255-415, 588-637).  Zeroes the blocks it is about to fill. */
static int decode_mcus(const struct jpeg_hdr *j, int16_t *const planes[3], struct bits *b, long mcu, long count)
{
    int pred[3] = {0, 0, 0};
    for (int c = 0; c < j->ncomp; c++) {
        const size_t per = (size_t)j->h[c] * j->v[c] * 64;
        memset(planes[c] + (size_t)mcu * per, 0, (size_t)count * per * sizeof(int16_t));
    }
    for (const long end = mcu + count; mcu < end; mcu++) {
        for (int c = 0; c < j->ncomp; c++) {
            const struct huff *hd = &j->dc[j->td[c]], *ha = &j->ac[j->ta[c]];
            const int nb = j->h[c] * j->v[c];
            for (int k = 0; k < nb; k++) {
                int16_t *blk = planes[c] + (mcu * nb + k) * 64;
                const int t = huff_decode(b, hd);
                if (t < 0 || t > 11) return FFHIP_EINVAL;
                pred[c] += extend(bits_get(b, t), t);
                blk[0] = (int16_t)pred[c];
                for (int i = 1; i < 64;) {
                    if (b->n < 16) bits_fill(b);
                    const int f = ha->fast[(b->acc >> (b->n - LOOK)) & ((1u << LOOK) - 1)];
                    if (f) { /* run, value and all their bits from one look-up */
                        i += (f >> 4) & 15;
                        if (i > 63) return FFHIP_EINVAL;
                        b->n -= f & 15;
                        blk[k_zigzag[i++]] = (int16_t)(f >> 8);
                        continue;
                    }
                    const int rs = huff_decode(b, ha);
                    if (rs < 0) return FFHIP_EINVAL;
                    const int r = rs >> 4, s = rs & 15;
                    if (s == 0) {
                        if (r == 15) { i += 16; continue; }
                        break; /* EOB */
                    }
                    i += r;
                    if (i > 63) return FFHIP_EINVAL;
                    blk[k_zigzag[i]] = (int16_t)extend(bits_get(b, s), s);
                    i++;
                }
            }
        }
    }
    /* the bytes fed behind the end of the data are look-ahead; a stream that consumed any of them is truncated
     * (the reference's reader overruns its buffer there, utils/bitstream.c:117: nothing to be in parity with) */
    if (b->dry * 8 > b->n) return FFHIP_EINVAL;
    return FFHIP_OK;
}

/* restart intervals [first, last) of one picture; seg[i] = offset of interval i's first byte in the scan */
struct interval_job {
    const struct jpeg_hdr *j;
    int16_t *planes[3];
    const size_t *seg;
    long n_seg, mcus, first, last;
    int rc;
};
static void *interval_worker(void *arg)
{
    struct interval_job *w = arg;
    w->rc = FFHIP_OK;
    for (long i = w->first; i < w->last && w->rc == FFHIP_OK; i++) {
        const uint8_t *p = w->j->scan + w->seg[i];
        const uint8_t *e = i + 1 < w->n_seg ? w->j->scan + w->seg[i + 1] - 2 : w->j->scan + w->j->scan_len; /* stop at the RSTn */
        struct bits b = {p, e, 0, 0, 0, 0};
        const long mcu = i * w->j->restart, left = w->mcus - mcu;
        w->rc = decode_mcus(w->j, w->planes, &b, mcu, left < w->j->restart ? left : w->j->restart);
    }
    return NULL;
}

/* one picture: coefficient planes (MCU order, natural order inside a block) + quant tables.
 * n_threads > 1 and a DRI segment in the file: the restart intervals (independent by construction,
 * jpg.c:562-573) are shared out over host threads. */
int ffhip_jpeg_entropy_decode_mt(const uint8_t *file, size_t len, const ffhip_jpeg_geom *expect, int16_t *coef_y,
                                 int16_t *coef_u, int16_t *coef_v, uint16_t *quant /* [4][64] */, int n_threads)
{
    struct jpeg_hdr *j = malloc(sizeof *j);
    if (!j) return FFHIP_ENOMEM;
    int rc = file && coef_y && quant ? parse_headers(file, len, j) : FFHIP_EINVAL;
    if (rc) { free(j); return rc; }
    const int mcu_cols = (j->width + 8 * j->h[0] - 1) / (8 * j->h[0]), mcu_rows = (j->height + 8 * j->v[0] - 1) / (8 * j->v[0]);
    if (expect && (expect->mcu_cols != mcu_cols || expect->mcu_rows != mcu_rows || expect->ncomp != j->ncomp ||
                   expect->h != j->h[0] || expect->v != j->v[0])) { free(j); return FFHIP_EINVAL; }
    if (j->ncomp == 3 && (!coef_u || !coef_v)) { free(j); return FFHIP_EINVAL; }
    memcpy(quant, j->quant, sizeof j->quant);
    const long mcus = (long)mcu_cols * mcu_rows;
    struct interval_job base = {j, {coef_y, coef_u, coef_v}, NULL, 1, mcus, 0, 1, FFHIP_OK};
    if (!j->restart) { /* one segment, one thread */
        struct bits b = {j->scan, j->scan + j->scan_len, 0, 0, 0, 0};
        rc = decode_mcus(j, base.planes, &b, 0, mcus);
        free(j);
        return rc;
    }
    /* where every restart interval starts: behind the RSTn markers (0xFF is stuffed inside entropy data,
     * so FF D0..D7 can only be a marker) */
    const long n_seg = (mcus + j->restart - 1) / j->restart;
    size_t *seg = malloc((size_t)n_seg * sizeof *seg);
    if (!seg) { free(j); return FFHIP_ENOMEM; }
    long found = 1;
    seg[0] = 0;
    for (size_t q = 0; q + 1 < j->scan_len && found < n_seg; q++)
        if (j->scan[q] == 0xFF && j->scan[q + 1] >= 0xD0 && j->scan[q + 1] <= 0xD7) { seg[found++] = q + 2; q++; }
    if (found != n_seg) { free(seg); free(j); return FFHIP_EINVAL; } /* a marker is missing: the reference would run dry too */
    base.seg = seg;
    base.n_seg = n_seg;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n_seg) n_threads = (int)n_seg;
    if (n_threads > 64) n_threads = 64;
    struct interval_job jobs[64];
    pthread_t tid[64];
    int started[64] = {0};
    for (int t = 0; t < n_threads; t++) {
        jobs[t] = base;
        jobs[t].first = n_seg * t / n_threads;
        jobs[t].last = n_seg * (t + 1) / n_threads;
        if (t) started[t] = pthread_create(&tid[t], NULL, interval_worker, &jobs[t]) == 0;
    }
    interval_worker(&jobs[0]);
    rc = jobs[0].rc;
    for (int t = 1; t < n_threads; t++) {
        if (started[t]) pthread_join(tid[t], NULL);
        else interval_worker(&jobs[t]); /* could not start a thread: do its share here */
        if (rc == FFHIP_OK) rc = jobs[t].rc;
    }
    free(seg);
    free(j);
    return rc;
}

int ffhip_jpeg_entropy_decode(const uint8_t *file, size_t len, const ffhip_jpeg_geom *expect, int16_t *coef_y,
                              int16_t *coef_u, int16_t *coef_v, uint16_t *quant /* [4][64] */)
{
    return ffhip_jpeg_entropy_decode_mt(file, len, expect, coef_y, coef_u, coef_v, quant, 1);
}

struct batch_job {
    const uint8_t *const *files;
    const size_t *lens;
    const ffhip_jpeg_geom *g;
    int first, last;
    int16_t *y, *u, *v;
    uint16_t *quant;
    int *status;
};
/* a picture that failed to parse or decode still occupies its place in the batch: its planes must not keep what an
 * earlier chunk or call left in the (reused, pinned) buffers -- all-zero coefficients reconstruct to a flat picture */
static void blank_picture(int16_t *y, int16_t *u, int16_t *v, uint16_t *quant, size_t yb, size_t cb)
{
    memset(y, 0, yb * sizeof(int16_t));
    if (u) memset(u, 0, cb * sizeof(int16_t));
    if (v) memset(v, 0, cb * sizeof(int16_t));
    for (int i = 0; i < 256; i++) quant[i] = 1;
}
static void *batch_worker(void *arg)
{
    struct batch_job *jb = arg;
    const size_t mcus = (size_t)jb->g->mcu_cols * jb->g->mcu_rows, yb = mcus * jb->g->h * jb->g->v * 64, cb = mcus * 64;
    for (int i = jb->first; i < jb->last; i++) {
        jb->status[i] = ffhip_jpeg_entropy_decode(jb->files[i], jb->lens[i], jb->g, jb->y + i * yb, jb->u ? jb->u + i * cb : NULL,
                                                  jb->v ? jb->v + i * cb : NULL, jb->quant + (size_t)i * 256);
        if (jb->status[i]) blank_picture(jb->y + i * yb, jb->u ? jb->u + i * cb : NULL, jb->v ? jb->v + i * cb : NULL, jb->quant + (size_t)i * 256, yb, cb);
    }
    return NULL;
}

/* n pictures of one geometry, statically partitioned over n_threads host threads; planes are
 * image-major exactly as ffhip_jpeg_recon_batch reads them (quant_stride 256). */
int ffhip_jpeg_entropy_batch(const uint8_t *const *files, const size_t *lens, int n, int n_threads,
                             const ffhip_jpeg_geom *geom, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v,
                             uint16_t *quant, int *status)
{
    if (n < 0 || !geom || (n > 0 && (!files || !lens || !coef_y || !quant || !status))) return FFHIP_EINVAL;
    if (n == 0) return FFHIP_OK;
    if (n_threads < 1) n_threads = 1;
    if (n_threads >= 2 * n) { /* more threads than pictures: spend them inside each picture (restart intervals) */
        const size_t mcus = (size_t)geom->mcu_cols * geom->mcu_rows, yb = mcus * geom->h * geom->v * 64, cb = mcus * 64;
        int first_err = FFHIP_OK;
        for (int i = 0; i < n; i++) {
            status[i] = ffhip_jpeg_entropy_decode_mt(files[i], lens[i], geom, coef_y + (size_t)i * yb, coef_u ? coef_u + (size_t)i * cb : NULL,
                                                     coef_v ? coef_v + (size_t)i * cb : NULL, quant + (size_t)i * 256, n_threads);
            if (status[i]) blank_picture(coef_y + (size_t)i * yb, coef_u ? coef_u + (size_t)i * cb : NULL, coef_v ? coef_v + (size_t)i * cb : NULL, quant + (size_t)i * 256, yb, cb);
            if (status[i] && !first_err) first_err = status[i];
        }
        return first_err;
    }
    if (n_threads > n) n_threads = n;
    if (n_threads > 256) n_threads = 256;
    struct batch_job jobs[256];
    pthread_t tid[256];
    int started[256] = {0};
    for (int t = 0; t < n_threads; t++) {
        jobs[t] = (struct batch_job){files, lens, geom, (int)((long)n * t / n_threads), (int)((long)n * (t + 1) / n_threads),
                                     coef_y, coef_u, coef_v, quant, status};
        if (t) started[t] = pthread_create(&tid[t], NULL, batch_worker, &jobs[t]) == 0;
    }
    batch_worker(&jobs[0]);
    for (int t = 1; t < n_threads; t++) {
        if (started[t]) pthread_join(tid[t], NULL);
        else batch_worker(&jobs[t]); /* could not start a thread: do its share here */
    }
    for (int i = 0; i < n; i++)
        if (status[i]) return status[i];
    return FFHIP_OK;
}

/* display/bmpwriter.c:19-81: 54-byte header (BITMAPINFOHEADER, negative height = top-down,
 * 32 bpp, the same odd constants 0x60 / biClrUsed 2) followed by the BGRA rows, tightly packed */
int ffhip_bmp_write(const char *path, const uint8_t *bgra, int width, int height, int64_t pitch)
{
    if (!path || !bgra || width <= 0 || height <= 0 || pitch < (int64_t)width * 4) return FFHIP_EINVAL;
    FILE *f = fopen(path, "wb");
    if (!f) return FFHIP_EIO;
    uint8_t h[54] = {0};
    const uint32_t img = (uint32_t)width * (uint32_t)height * 4u, size = 54u + img;
    const int32_t neg_h = -height;
#define PUT32(o, v) do { uint32_t v__ = (uint32_t)(v); h[o] = v__ & 255; h[(o) + 1] = (v__ >> 8) & 255; h[(o) + 2] = (v__ >> 16) & 255; h[(o) + 3] = v__ >> 24; } while (0)
    h[0] = 'B'; h[1] = 'M';
    PUT32(2, size); PUT32(10, 0x36); PUT32(14, 0x28); PUT32(18, width); PUT32(22, neg_h);
    h[26] = 1; h[28] = 32;
    PUT32(34, img); PUT32(38, 0x60); PUT32(42, 0x60); PUT32(46, 2);
#undef PUT32
    int ok = fwrite(h, 54, 1, f) == 1;
    for (int y = 0; ok && y < height; y++) ok = fwrite(bgra + (size_t)y * (size_t)pitch, (size_t)width * 4, 1, f) == 1;
    ok = (fclose(f) == 0) && ok;
    return ok ? FFHIP_OK : FFHIP_EIO;
}
