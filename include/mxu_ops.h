/*
 * mxu_ops.h -- the reference's directly callable compute kernels, served by
 * the GPU.  Prototypes of reference include/mxu_ops.h:21-82 plus
 * conv2d_int8_nhwc_mxu (declared only in reference mars_runtime.c:31-38).
 *
 * All pointers are HOST pointers, as in the reference; each call stages its
 * operands to HBM, runs the HIP kernel and copies the result back
 * (synchronous).  The device-pointer forms used by the graph executor are in
 * mars_hip.h.  Layouts (note: the comments in the reference header say
 * NHWC/OHWI for all three, the code is as stated here):
 *   conv2d_int8_mxu       in [C,H,W]  w [O,I,kh,kw]  out [O,H,W]  (mxu_conv.c:630-670)
 *   conv2d_int8_nhwc_mxu  in [H,W,C]  w [O,kh,kw,I]  out [H,W,O]  (mxu_conv.c:713-757)
 *   conv2d_float32_mxu    in [C,H,W]  w [O,I,kh,kw]  out [O,H,W]  (mxu_conv.c:673-710)
 * Requantisation: r = (int32)(acc*cs +/- 0.5f) with cs = (in_scale*w_scale)/out_scale
 * in float32, x86 truncation semantics (out of range / NaN -> INT_MIN), clamp
 * to [-128,127].
 */
#ifndef MXU_OPS_H
#define MXU_OPS_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Kept for source compatibility (reference mxu_ops.c:29-61).  mxu_init marks
 * the flag; there is no coprocessor to enable. */
void mxu_init(void *nna_mem);
int mxu_is_initialized(void);

void mxu_mul_f32(float *out, const float *a, const float *b, size_t count);
void mxu_add_f32(float *out, const float *a, const float *b, size_t count);
void mxu_sub_f32(float *out, const float *a, const float *b, size_t count);
void mxu_relu_f32(float *out, const float *in, size_t count);

void conv2d_int8_mxu(const signed char *input, int in_h, int in_w, int in_c,
                     const signed char *weight, int out_c, int kh, int kw,
                     const int *bias, signed char *output, int out_h, int out_w,
                     int stride_h, int stride_w, int pad_top, int pad_left,
                     float in_scale, float w_scale, float out_scale);

void conv2d_int8_nhwc_mxu(const signed char *input, int in_h, int in_w, int in_c,
                          const signed char *weight, int out_c, int kh, int kw,
                          const int *bias, signed char *output, int out_h, int out_w,
                          int stride_h, int stride_w, int pad_top, int pad_left,
                          float in_scale, float w_scale, float out_scale);

void conv2d_float32_mxu(const float *input, int in_h, int in_w, int in_c,
                        const float *weight, int out_c, int kh, int kw,
                        const float *bias, float *output, int out_h, int out_w,
                        int stride_h, int stride_w, int pad_top, int pad_left,
                        float *scratch);

#ifdef __cplusplus
}
#endif
#endif
