/* ffhip_entropy_internal.h -- what ffhip_entropy.c shares with the GPU entropy decoder (ffhip_huff_gpu.hip):
 * the parsed JPEG header with its Huffman look-up tables.  Internal to libffpic_hip.so. */
#ifndef FFHIP_ENTROPY_INTERNAL_H
#define FFHIP_ENTROPY_INTERNAL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LOOK 9
struct huff {
    uint16_t look[1 << LOOK]; /* (length << 8) | symbol, 0 = not resolvable in LOOK bits */
    int32_t maxcode[18];      /* per length, -1 if none */
    int32_t valptr[17], mincode[17];
    uint8_t vals[256];
    int present;
    /* AC tables: when code and magnitude bits both fit into LOOK bits, the whole coefficient in one
     * look-up: (value << 8) | (run << 4) | (code length + magnitude bits), 0 = take the slow path */
    int16_t fast[1 << LOOK];
};

struct jpeg_hdr {
    int width, height, ncomp, restart;
    int h[3], v[3], tq[3], td[3], ta[3], cid[3];
    uint16_t quant[4][64];
    struct huff dc[4], ac[4];
    const uint8_t *scan;
    size_t scan_len;
};

/* marker loop, SOF/DQT/DHT/DRI/SOS parsing (format/jpg.c:78-105, 640-655, 771-855); 0 or FFHIP_EINVAL */
int ffhip_jpeg_parse(const uint8_t *file, size_t len, struct jpeg_hdr *j);
int ffhip_jpeg_probe_restart(const uint8_t *file, size_t len); /* DRI value, 0 = none, -1 = does not parse */

#ifdef __cplusplus
}
#endif
#endif
