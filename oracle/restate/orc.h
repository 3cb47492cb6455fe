/*
 * orc.h -- CPU restatement of the reference's .mars hot path ("the oracle").
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may call into this library; it is the checker,
 * never the product and never a fallback for it.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit against
 * the reference's own sources compiled in place (oracle/_ref, see
 * tests/test_oracle_vs_ref.py) and against the golden vectors those produced
 * (tests/golden/, generator tests/golden/make_golden.py).
 *
 * Memory semantics are those of oracle O2: the weight
 * blob is followed by zeros, and every activation tensor lives alone in its
 * own zero-initialised buffer with zero slack behind it.
 */
#ifndef ORC_H
#define ORC_H

#include <stddef.h>
#include <stdint.h>

#include "mars.h"

#ifdef __cplusplus
extern "C" {
#endif

/* float -> int32 exactly as x86 cvttss2si does it: truncate toward zero,
 * anything unrepresentable (|x| too large, NaN) becomes INT32_MIN.
 * SURVEY.md appendix B.2. */
int32_t orc_trunc_x86(float x);

/* ---- convolution kernels (reference src/mars/mxu_conv.c, portable branch) */
typedef struct {
    int in_h, in_w, in_c;
    int out_h, out_w, out_c;
    int kh, kw;
    int stride_h, stride_w;
    int pad_top, pad_left;
} orc_conv_geom_t;

/* mxu_conv.c:713-757 -- in [H,W,C], w [O,kh,kw,I], out [H,W,O] */
void orc_conv_i8_nhwc(const int8_t *in, const int8_t *w, const int32_t *bias, int8_t *out,
                      const orc_conv_geom_t *g, float in_scale, float w_scale, float out_scale);
/* mxu_conv.c:630-670 -- in [C,H,W], w [O,I,kh,kw], out [O,H,W] */
void orc_conv_i8_nchw(const int8_t *in, const int8_t *w, const int32_t *bias, int8_t *out,
                      const orc_conv_geom_t *g, float in_scale, float w_scale, float out_scale);
/* mxu_conv.c:673-710 -- f32, NCHW/OIHW, sequential (ic,kh,kw) accumulation */
void orc_conv_f32_nchw(const float *in, const float *w, const float *bias, float *out,
                       const orc_conv_geom_t *g);

/* ---- element-wise / data movement (reference src/mars/mars_runtime.c) */
void orc_relu_bytes(int8_t *buf, size_t n);                                   /* :700-707 */
void orc_sigmoid_i8(const int8_t *in, int8_t *out, size_t n, float in_scale, float out_scale); /* :752-768 */
void orc_sigmoid_f32(const float *in, float *out, size_t n);                  /* :742-749 */
void orc_binary_i8(int is_mul, const int8_t *a, const int8_t *b, int8_t *out, size_t n,
                   float sa, float sb, float so);                             /* :818-835, :885-902 */
void orc_binary_f32(int is_mul, const float *a, const float *b, float *out, size_t n); /* :807-816, :874-883 */
void orc_relu_i8(const int8_t *in, int8_t *out, size_t n, int leaky);         /* :1072-1086 */
void orc_relu_f32(const float *in, float *out, size_t n, int leaky);          /* :1066-1071 */
void orc_maxpool_i8(const int8_t *in, int8_t *out, int in_h, int in_w, int ch, int out_h,
                    int out_w, int kh, int kw, int sh, int sw);               /* :919-957 */
void orc_concat_slice_i8(const int8_t *in, int8_t *out, int out_h, int out_w, int in_c,
                         int out_c, int ch_off);                              /* :982-996 */
void orc_upsample_i8(const int8_t *in, int8_t *out, int in_h, int in_w, int ch, int out_h,
                     int out_w, int scale_h, int scale_w);                    /* :1026-1041 */
void orc_batchnorm_i8(const int8_t *in, int8_t *out, int n, int c, int h, int w, const float *s,
                      const float *b, float in_scale, float out_scale);       /* :1131-1154 */
void orc_batchnorm_f32(const float *in, float *out, int n, int c, int h, int w, const float *s,
                       const float *b);                                       /* :1115-1130 */

/* ---- detection tail (reference src/mars/mars_yolo_test.c:80-130) */
typedef struct {
    float x, y, w, h, conf;
    int cls;
} orc_det_t;
int orc_parse_output(const int8_t *pred, int npred, float scale, orc_det_t *dets, int maxd);
int orc_nms(orc_det_t *d, int n, float thresh);

/* ---- whole graph, O2 memory semantics */
typedef struct orc_graph orc_graph_t;
/* returns NULL on malformed file; *err gets a mars_error_t-compatible code */
orc_graph_t *orc_graph_open(const void *file, size_t size, size_t slack_mult, size_t slack_add, int *err);
int orc_graph_num_tensors(const orc_graph_t *g);
int orc_graph_num_layers(const orc_graph_t *g);
int orc_graph_input_id(const orc_graph_t *g, int i);  /* tensor index of graph input i, -1 if none */
int orc_graph_output_id(const orc_graph_t *g, int i);
size_t orc_graph_tensor_bytes(const orc_graph_t *g, int tensor_index); /* numel * elemsize by shape */
void *orc_graph_tensor(orc_graph_t *g, int tensor_index, size_t *alloc);
int orc_graph_set_input(orc_graph_t *g, int input_index, const void *data, size_t bytes);
int orc_graph_run_range(orc_graph_t *g, int first, int last); /* mars_error_t of first failing layer */
int orc_graph_run(orc_graph_t *g);
void orc_graph_zero_activations(orc_graph_t *g);
void orc_graph_close(orc_graph_t *g);
double orc_graph_conv_macs(const orc_graph_t *g); /* sum over CONV2D layers, per frame */

/* Run `nframes` independent frames (frame-major input/output 0) on `nthreads`
 * host threads, one private graph instance per thread.  Used by the
 * cpu_baseline leg and by batch-parity tests.  Returns 0 or a mars_error_t. */
int orc_run_frames(const void *file, size_t size, const void *inputs, size_t in_stride,
                   void *outputs, size_t out_stride, int out_index, int nframes, int nthreads);

/* ---- image front-end (reference src/mars/mars_yolo_test.c:40-77 load_image, minus the file decode) */
/* stbir_resize_uint8(in, w, h, 0, out, ow, oh, 0, 3) of the vendored stb_image_resize.h, restated */
int orc_resize_rgb8(const uint8_t *in, int w, int h, uint8_t *out, int ow, int oh);
/* RGB uint8 [h][w][3] -> letterboxed int8 (px - 128, pad -17) [th][tw][3] (nhwc) or [3][th][tw] */
int orc_letterbox(const uint8_t *rgb, int w, int h, int tw, int th, int nhwc, int8_t *out);

#ifdef __cplusplus
}
#endif
#endif
