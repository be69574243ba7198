/* ffhip_internal.h -- shared by the .hip translation units of libffpic_hip.so. */
#ifndef FFHIP_INTERNAL_H
#define FFHIP_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ffpic_hip.h"

#define FFHIP_CHECK(expr, code)                    \
    do {                                           \
        hipError_t e__ = (expr);                   \
        if (e__ != hipSuccess) {                   \
            ffhip_note_hip_error((int)e__, #expr); \
            return (code);                         \
        }                                          \
    } while (0)

#ifdef __cplusplus
extern "C" {
#endif
void ffhip_note_hip_error(int hip_error, const char *what);
int ffhip_have_device(void); /* 1 once ffhip_init succeeded on a gfx950 device */
int *ffhip_async_err_word(void); /* pinned word kernels report an in-launch abort through; see ffhip_stream_sync */
#ifdef __cplusplus
}
#endif

typedef unsigned int u32;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#endif
