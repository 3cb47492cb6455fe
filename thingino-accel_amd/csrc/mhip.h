/*
 * mhip.h -- the thin C-ABI layer between the C host code (csrc/host) and the
 * HIP translation units (csrc/hip).  Plain pointers, sizes and small POD
 * structs only; no HIP types cross this boundary.  Every launcher enqueues on
 * the library's single stream and returns 0 or a negative hipError_t.
 *
 * Replaces, conceptually: ioctl/mmap device access (reference src/device.c),
 * NNDMA staging (reference src/nna_dma.c) and the MXUv3 kernels (reference
 * src/mars/mxu_conv.c, mxu_ops.c).
 */
#ifndef MHIP_H
#define MHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime */
int mhip_init(int device_hint);          /* 0 ok; <0: no device / wrong arch */
void mhip_shutdown(void);
int mhip_ready(void);
int mhip_device_info(int *cu_count, int *lds_bytes, int *gfx_version, size_t *hbm_bytes);
void *mhip_stream(void);
int mhip_sync(void);                    /* both streams */
void mhip_select_aux(int on);          /* launchers enqueue on the auxiliary stream while on */
void mhip_select_stream(int which);    /* ... or on: 0 main, 1 auxiliary, 2 upload, 3 download (pipelined I/O), 4 second compute */
int mhip_stream_wait(int which, void *ev); /* that stream waits for an event */
int mhip_event_sync(void *ev);         /* host waits for an event */
void *mhip_malloc(size_t bytes);         /* HBM */
void mhip_free(void *p);
void *mhip_host_alloc(size_t bytes);     /* pinned + device mapped */
void mhip_host_free(void *p);
int mhip_memset_async(void *dst, int value, size_t bytes);
int mhip_h2d_async(void *dst, const void *src, size_t bytes);
int mhip_d2h_async(void *dst, const void *src, size_t bytes);
int mhip_d2d_async(void *dst, const void *src, size_t bytes);
/* strided copies: `rows` rows of `row_bytes`, source/destination pitches */
int mhip_h2d_2d_async(void *dst, size_t dpitch, const void *src, size_t spitch, size_t row_bytes, size_t rows);
int mhip_d2h_2d_async(void *dst, size_t dpitch, const void *src, size_t spitch, size_t row_bytes, size_t rows);
/* event pairs for per-op timing */
void *mhip_event_create(void);
void *mhip_event_create_sync(void); /* ordering only (no timing): for stream -> stream hand-offs */
void mhip_event_destroy(void *ev);
int mhip_event_record(void *ev);
float mhip_event_elapsed_ms(void *start, void *stop); /* waits for stop */
const char *mhip_last_error(void);
/* capture the launches enqueued on the main stream between begin and end into an executable graph (NULL on failure) */
int mhip_graph_begin(void);
void *mhip_graph_end(int ok);
int mhip_graph_launch(void *exec);
void mhip_graph_destroy(void *exec);

/* ---- int8 convolution (conv_i8.hip) */
typedef struct {
    const int8_t *in;  size_t in_stride;   /* per-frame stride, bytes */
    int8_t *out;       size_t out_stride;
    const int8_t *w;   /* packed: [oc_pad][kh][row_pad], see mhip_conv_i8_pack_geom */
    const int32_t *bias; /* [oc_pad] or NULL */
    const uint8_t *lut;  /* 256-entry post-requant map (index q+128) or NULL */
    const uint8_t *lut2; /* optional 512-entry half-step form of `lut` (index trunc(2*acc*cs)+256, lower clamp folded in);
                            only valid when mhip_conv_i8_lut2_ok(cs): enables the 4-instruction requantisation */
    /* optional: a 1x1 stride-1 convolution with a fused SiLU table (in_c -> in_c channels) evaluated on the staged input
     * patch BEFORE this k x k convolution reads it (C3 bottleneck: cv2(cv1(x)), patch-staged kernel only):
     * packed weights [in_c][64], bias rows, half-step table, combined scale */
    const int8_t *pre_w;
    const int32_t *pre_bias;
    const uint8_t *pre_lut2;
    float pre_cs;
    const int8_t *w_rgb;        /* optional: the RGB stem's A operands in the layout conv_i8_rgb keeps in LDS
                                 * (mhip_conv_i8_rgb_pack); NULL = the kernel re-lays p.w itself, once per workgroup */
    const int8_t *w_rows;       /* optional: the weights as conv_i8_rows streams them (mhip_conv_i8_rows_pack): one 8 KB LDS
                                 * image per (channel tile, 64-channel chunk, tap); NULL = that launch variant is not offered */
    int frames;
    int in_h, in_w, in_c;       /* input as NHWC */
    int out_h, out_w, out_c;
    int kh, kw, stride_h, stride_w, pad_top, pad_left;
    int row_pad;                /* bytes per kernel row in the packed weights (multiple of 16) */
    int oc_pad;                 /* packed output channels (multiple of 16) */
    float cs;                   /* (in_scale*w_scale)/out_scale, evaluated on the host in f32 */
    int relu;                   /* clamp negative results to 0 (fused ReLU, byte semantics) */
    int out_nchw;               /* store [O][H][W] instead of [H][W][O] */
    int in_planar;              /* > 0 (small-channel stem only, in_c == 4): `in` is in_planar (<= 4) PLANES [c][in_h][in_w] per frame -- an NCHW-tagged graph's
                                   input as the reference holds it -- which conv_i8_smallc interleaves while it stages its patch (round 6: no relayout launch);
                                   mhip_conv_i8 returns -2 when that kernel does not take the shape (the caller relays the input and launches again) */
    int safe;                   /* host proved |acc*cs| < 2^31 and cs finite: skip the x86 overflow fix-up */
    int out_pix_stride;         /* NHWC only: bytes between consecutive output pixels (0 = out_c); with out_ch_off
                                   this writes straight into a channel slice of a wider tensor (zero-copy concat) */
    int out_ch_off;
    int variant;                /* 0 = default launch policy, else a code from mhip_conv_i8_variants() */
    /* virtual concatenation (1x1 convolutions): when nseg > 1 the input pixel's in_c channels are the
     * concatenation of nseg dense NHWC tensors; `in` is unused.  seg_c0 = first channel of a segment (unused
     * entries 0x7fffffff), every seg_c a multiple of 32.  See mhip_conv_i8_seg_ok(). */
    /* fused residual Add (ADD layer folded into this convolution's epilogue): `add` = the other operand, an int8
     * tensor with exactly the output's layout and frame stride (NULL = none); out = sat8(trunc((conv*add_s_conv +
     * other*add_s_other)*add_inv + 0.5f)).  Only the one-tile and patch-staged forms implement it. */
    const int8_t *add;
    float add_s_conv, add_s_other, add_inv;
    int nseg;
    const int8_t *seg_in[4];
    size_t seg_stride[4];
    int seg_c[4], seg_c0[4];
    int seg_up;                 /* bit k: segment k is read through a 2x2 nearest upsample (its tensor is out_h/2 x out_w/2) */
} mhip_conv_i8_t;
/* row of the packed weights / bias that holds output channel oc (channels are permuted so that a lane's
 * results are consecutive channels) */
int mhip_conv_i8_oc_row(int oc, int oc_pad);
/* Bytes of, and (out != NULL) the content of, conv_i8_rgb's LDS weight image for this geometry; 0 = not an RGB-stem shape
 * that kernel takes.  `packed` = the layer's packed weights [oc_pad][k64] (mars_pack_conv_i8, 4-byte taps). */
/* can the convolution described by *p take a preceding 1x1 (pre_* fields) in its launch?  (geometry only) */
int mhip_conv_i8_pre_ok(const mhip_conv_i8_t *p);
size_t mhip_conv_i8_rgb_pack(int in_c, int kh, int kw, int stride_h, int stride_w, int pad_left, int oc_pad, int k64,
                             const int8_t *packed, int8_t *out);
/* Bytes of, and (out != NULL) the content of, conv_i8_rows' weight image (deep 3x3 stride-1 layers); 0 = not a shape that
 * kernel takes (out_w = the map width, 0 = unknown / any) */
size_t mhip_conv_i8_rows_pack(int in_c, int kh, int kw, int stride_h, int stride_w, int oc_pad, int k64, int out_w,
                              const int8_t *packed, int8_t *out);
unsigned long mhip_conv_i8_rows_launches(void); /* launches of conv_i8_rows since load (diagnostic) */
/* is the float->int conversion of acc*cs provably in range for every int32 accumulator? */
int mhip_conv_i8_is_safe(float cs);
/* may the half-step LUT be used for this combined scale?  (no int32 accumulator may requantise to +-0x3EFFFFFF) */
int mhip_conv_i8_lut2_ok(float cs);
int mhip_conv_i8_tune(const char *key, int value); /* launch-policy knobs, see mars_hip_set_tuning */
int mhip_conv_i8_tune_get(const char *key, int *value);
/* launch variants that can run this layer (same bytes out, different speed), the default first; 0 if none */
int mhip_conv_i8_variants(const mhip_conv_i8_t *p, int *codes, int max);
/* can this (shape-only: pointers may be dummies) segmented convolution run?  frames/out_stride as they will be */
int mhip_conv_i8_seg_ok(const mhip_conv_i8_t *p);
/* two convolutions over the same input in one launch (the second's input reads hit L2); -2 = not eligible as a
 * pair (launch them separately), else the launch result */
int mhip_conv_i8_pair(const mhip_conv_i8_t *a, const mhip_conv_i8_t *b);
/* does a layer with in_c bytes per pixel take the small-channel packing (every pixel widened to 4 bytes, 32-byte kernel rows)? */
int mhip_conv_i8_small_c(int in_c, int kw, int out_c);
/* packing geometry shared by host packer and kernel */
/* c_eff: bytes per input pixel in the packed K layout (4 in small-channel mode, else in_c) */
void mhip_conv_i8_pack_geom(int in_c, int kw, int out_c, int *row_pad, int *oc_pad, int *c_eff);
int mhip_conv_i8(const mhip_conv_i8_t *p);

/* ---- float32 convolution (conv_f32.hip): NCHW / OIHW, reference summation order */
typedef struct {
    const float *in;  size_t in_stride;
    float *out;       size_t out_stride;
    const float *w;   const float *bias;
    const void *w_split; /* optional: the weights cut into bf16 planes (mhip_conv_f32_split_pack): conv_f32_split needs it */
    const void *w_patch; /* optional: unit table, schedule and the weights' two bf16 planes in conv_f32_patch's K order
                            (mhip_conv_f32_patch_pack): k x k layers take that kernel under use_mfma == 3 when it is there */
    int frames;
    int in_h, in_w, in_c, out_h, out_w, out_c;
    int kh, kw, stride_h, stride_w, pad_top, pad_left;
    int silu;     /* fused conv -> SIGMOID -> MUL chain (float forms): out = v * (1 / (1 + expf(-v))), libm-exact expf */
    const float *add; size_t add_stride; /* optional fused residual Add (reference mars_runtime.c:807-816, the float ADD): out = result + add[same index],
                                            one float add after the SiLU -- the same float the separate layer computes */
    int use_mfma; /* 0: reference summation order, bit-identical; 1: implicit GEMM on v_mfma_f32_16x16x4_f32 (fused
                     rounding per tap: inside the 1e-4 tolerance of the float32 models, not bit-equal); 2: implicit GEMM on
                     v_mfma_f32_16x16x32_bf16 with every operand split into three bf16 pieces, six piece products per
                     product (conv_f32_split.hip: errors of the size of an f32 rounding); 3: the same with two pieces
                     and three piece products ("bf16x3": relative error per product <= 2^-16, same tolerance class) */
    int k_limit;  /* use_mfma >= 2 (conv_f32_split): input channels >= k_limit are known to be exact zeros in every frame (the planner proves it:
                     mars_plan.c zero_tail_f32 -- the reference's byte-wise CONCAT writes a quarter of a float tensor's bytes): the K loop stops
                     there.  Adding +-0 to an f32 accumulator changes nothing but the sign of a zero sum.  0 = no such knowledge */
    int k_limit_required; /* the planes from k_limit on are NOT zeros but memory the launch must not sum (`in` is a shifted view of another
                     tensor: mars_plan.c virtual_concat_f32): only a kernel that honours k_limit may run it -- the launch fails otherwise */
    int in_rec, out_rec; /* use_mfma == 3 only: the input (1: read by conv_f32_prec, 2: by conv_f32_patch's record-input form; mhip_conv_f32_patch_rec_form) / the
                            output is in RECORD format instead of NCHW floats -- [c / 8][h][w] records of
                            32 bytes = [8 x bf16 hi | 8 x bf16 mid] of 8 consecutive channels of one pixel (hi = bf16(x), mid = bf16(x - hi):
                            the two pieces conv_f32_patch cuts every input into anyway).  Same bytes per element; a tensor written by one
                            convolution for ONE k x k convolution to read (the planner pairs them: mars_plan.c rec_pairs).  in_rec needs
                            w_patch packed by mhip_conv_f32_patch_pack2(rec = 1) (conv_f32_prec); out_rec: conv_f32_split / conv_f32_stem,
                            no fused Add, out_c a multiple of 8 */
} mhip_conv_f32_t;
int mhip_conv_f32(const mhip_conv_f32_t *p);
/* the first pixels of a 1 x 1 convolution over a never-materialised byte-wise CONCAT of float maps (conv_f32_vcat.hip): p->in = the concat's LAST
 * input, p->k_limit = planes to sum, w_t = the weights of those planes transposed to [plane][out_c] (mhip_conv_f32_vcat_pack: bytes, and with w and out
 * the content); first[n_first] = its other inputs in order, run_floats = floats per concat run (shape[3] / 4).  Overwrites pixels
 * [0, n_first * run_floats) of every output plane (or their records: out_rec) */
size_t mhip_conv_f32_vcat_pack(int out_c, int in_c, int planes, const float *w, float *out);
int mhip_conv_f32_vcat_head(const mhip_conv_f32_t *p, const float *w_t, const float *const *first, const size_t *first_strides, int n_first, int run_floats);
/* two float convolutions over the same input in one launch (use_mfma >= 2, conv_f32_split: same geometry and channel count, no residual Add, no
 * record operands -- C3's cv1 + cv2): the second one's input reads hit L2; -2 = not eligible as a pair (launch them separately) */
int mhip_conv_f32_pair(const mhip_conv_f32_t *a, const mhip_conv_f32_t *b);
unsigned long mhip_conv_f32_pair_launches(void); /* paired launches since load (diagnostic) */
/* Bytes of, and (out != NULL) the content of, the weight image conv_f32_split reads: `planes` (2: hi, mid; 3: hi, mid, lo) planes of bf16
 * [oc_pad][k_pad] -- oc_pad = roundup128(out_c), k_pad = roundup64(K') + 64, zero filled -- with w = hi + mid + lo exactly
 * (hi = bf16(w), mid = bf16(w - hi), round to nearest; use_mfma == 3 reads the first two planes).  K' = in_c * kh * kw, or in_c * kh * (kw + 1) for stride_w == 2 with an odd kw (one zero column
 * appended to every kernel row: taps come in pairs there). */
size_t mhip_conv_f32_split_pack(int out_c, int in_c, int kh, int kw, int stride_w, int planes, const float *w, void *out);
unsigned long mhip_conv_f32_split_launches(void); /* launches of conv_f32_split since load (diagnostic) */
/* which gather form conv_f32_split takes for this shape (1 / 2: 16-byte gathers at stride 1 / 2; 0: a dword per tap), or -1: it declines the
 * shape whatever the batch (a weight image alone does not make a layer one it runs: the planner asks before it plans record output) */
int mhip_conv_f32_split_takes(int out_c, int in_c, int kh, int kw, int stride_h, int stride_w, int pad_left, int in_w, int out_w);
/* conv_f32_patch (conv_f32_patch.hip: the input patch of a pixel tile staged and split once, k x k layers with >= 8 taps, stride
 * 1 / 2, in_c a multiple of 8 and >= 32, map widths multiples of 4).  Bytes of, and (w, out != NULL) the content of, its image:
 * [nsteps][4] unit offsets, [nsteps] chunk schedule, two bf16 planes [oc_pad][kp] in its K order; 0 = not such a shape */
size_t mhip_conv_f32_patch_pack(int out_c, int in_c, int kh, int kw, int stride, int pad, int in_h, int in_w, int out_h, int out_w,
                                const float *w, void *out);
/* rec != 0: the image for record-format input (conv_f32_prec: DMA issue schedule, a ring of up to four slots) */
size_t mhip_conv_f32_patch_pack2(int out_c, int in_c, int kh, int kw, int stride, int pad, int in_h, int in_w, int out_h, int out_w,
                                 int rec, const float *w, void *out);
int mhip_conv_f32_patch_geom2(int out_c, int in_c, int kh, int kw, int stride, int pad, int in_h, int in_w, int out_h, int out_w,
                              int rec, int *outv, int cap);
unsigned long mhip_conv_f32_prec_launches(void); /* launches of conv_f32_prec (record-format input) since load */
unsigned long mhip_conv_f32_recin_launches(void); /* ... of conv_f32_patch's record-input form (two slots, through registers) */
/* which kernel reads this layer's input when it arrives as records: 0 none, 1 conv_f32_prec (pack the image with rec = 1), 2 conv_f32_patch's
 * record-input form (the plain image) */
int mhip_conv_f32_patch_rec_form(int out_c, int in_c, int kh, int kw, int stride, int pad, int in_h, int in_w, int out_h, int out_w);
/* the layer geometry that kernel derives, as ints (tests, tools): see conv_f32_patch.hip; returns the count, 0 = not such a shape */
int mhip_conv_f32_patch_geom(int out_c, int in_c, int kh, int kw, int stride, int pad, int in_h, int in_w, int out_h, int out_w,
                             int *outv, int cap);
unsigned long mhip_conv_f32_patch_launches(void); /* launches of conv_f32_patch since load (diagnostic) */
/* conv_f32_stem (conv_f32_stem.hip: the float twins' first layer -- in_c <= 4, out_c <= 32, even kernel width, stride 2, output maps
 * multiples of 16 x 32): bytes of, and (w, out != NULL) the content of, the weight image it loads into registers (two bf16 planes
 * [32][20 units][8]); 0 = not such a shape.  The image travels in mhip_conv_f32_t.w_patch like conv_f32_patch's (a shape is taken by at
 * most one of the two kernels). */
size_t mhip_conv_f32_stem_pack(int out_c, int in_c, int kh, int kw, int stride, int pad, int in_h, int in_w, int out_h, int out_w,
                               const float *w, void *out);
unsigned long mhip_conv_f32_stem_launches(void);
/* policy knob: 0 never the matrix cores, 1 (default) wherever the host proves it safe, 2 everywhere, 3 / 4 everywhere on the
 * bf16 matrix cores with operands split in two / three (three / six piece products).  set < 0 only
 * reads; returns the mode in force (first call reads MARS_HIP_F32_MFMA) */
int mhip_conv_f32_mode(int set);

/* ---- element-wise (eltwise.hip).  n = elements per frame. */
int mhip_lut_i8(const int8_t *in, size_t in_stride, int8_t *out, size_t out_stride, int frames,
                size_t n, const uint8_t *lut_dev);
int mhip_relu_bytes(int8_t *buf, size_t stride, int frames, size_t n);
/* out_run > 0: the result is written as pixels of out_run channels into a wider tensor:
 * element i goes to (i / out_run) * out_pix_stride + out_ch_off + i % out_run (zero-copy concat) */
int mhip_binary_i8(int is_mul, const int8_t *a, size_t a_stride, const int8_t *b, size_t b_stride,
                   int8_t *out, size_t out_stride, int frames, size_t n, float sa, float sb, float inv_so,
                   int out_run, int out_pix_stride, int out_ch_off);
int mhip_sigmoid_f32(const float *in, size_t in_stride, float *out, size_t out_stride, int frames, size_t n);
int mhip_binary_f32(int op /*0 add,1 mul,2 sub*/, const float *a, size_t a_stride, const float *b,
                    size_t b_stride, float *out, size_t out_stride, int frames, size_t n);
int mhip_relu_f32(const float *in, size_t in_stride, float *out, size_t out_stride, int frames,
                  size_t n, float alpha);
int mhip_batchnorm_i8(const int8_t *in, size_t in_stride, int8_t *out, size_t out_stride, int frames,
                      int n, int c, int hw, const float *s, const float *b, float in_scale, float out_scale);
int mhip_batchnorm_f32(const float *in, size_t in_stride, float *out, size_t out_stride, int frames,
                       int n, int c, int hw, const float *s, const float *b);

/* ---- data movement (move.hip); all int8-byte semantics, NHWC index math */
/* out_pix_stride (0 = ch) / out_ch_off: as for the conv */
int mhip_maxpool_i8(const int8_t *in, size_t in_stride, int8_t *out, size_t out_stride, int frames,
                    int in_h, int in_w, int ch, int out_h, int out_w, int kh, int kw, int sh, int sw,
                    int out_pix_stride, int out_ch_off);
/* n <= 3 chained stride-1 max-pools (same window, output size = input size, ch % 16 == 0, h*w*64 <= 60 KB of LDS):
 * stage i reads stage i-1's result and writes outs[i] */
int mhip_pool_chain_i8(const int8_t *in, size_t in_stride, int8_t *const *outs, const size_t *out_strides, int n,
                       int frames, int h, int w, int ch, int kh, int kw);
int mhip_unpad_rows(const void *src, void *dst, size_t rows, int width, int pitch); /* device -> device */
int mhip_concat_slice(const int8_t *in, size_t in_stride, int8_t *out, size_t out_stride, int frames,
                      int out_h, int out_w, int in_c, int out_c, int ch_off);
/* the reference's CONCAT on [1, C, H, W]-tagged tensors of equal H, W (runs of W bytes, input n shifted n rows down, the last input wins), with
 * every operand held pixels x channels on the device; channel counts multiples of 16, n <= 4 (move.hip: concat_nchwq_kernel) */
int mhip_concat_nchwq(const int8_t *const *ins, const size_t *in_strides, const int *in_c, int n, int8_t *out, size_t out_stride,
                      int frames, int out_c, int H, int W, int rows_only /* > 0: only the first rows_only map rows of the output */);
/* ... and its UPSAMPLE (quirk dims qih x qiw x qch -> qoh x qow x qch, factors sh / sw) and stride-1, same-size MAXPOOL (window kh channels x kw
 * map rows) on such tensors: Ci / Co / C multiples of 16 */
int mhip_upsample_nchwq(const int8_t *in, size_t in_stride, int Ci, int Hi, int Wi, int8_t *out, size_t out_stride, int Co, int Ho, int Wo,
                        int frames, int qih, int qiw, int qch, int qoh, int qow, int sh, int sw);
int mhip_maxpool_nchwq(const int8_t *in, size_t in_stride, int8_t *out, size_t out_stride, int frames, int C, int H, int W, int kh, int kw);
int mhip_upsample_i8(const int8_t *in, size_t in_stride, int8_t *out, size_t out_stride, int frames,
                     int in_h, int in_w, int ch, int out_h, int out_w, int scale_h, int scale_w,
                     int out_pix_stride, int out_ch_off);
/* [C][HW] -> [HW][c_pad] with zero channel padding (feeds the NHWC conv kernel) */
int mhip_nchw_to_nhwc_pad(const int8_t *in, size_t in_stride, int8_t *out, size_t out_stride,
                          int frames, int c, int hw, int c_pad);

/* ---- detection tail (yolo_tail.hip) */
typedef struct {
    const int8_t *pred[4]; size_t stride[4]; int npred[4];
    int pix_c[4], pix_stride[4]; /* pix_stride != 0: every pix_c prediction bytes sit at the start of a pix_stride-byte pixel row */
    const float *lut[4];   /* per segment, device: 3 x 256 floats: value[q], obj[q], den[q] */
    int mono[4];           /* value[q] is strictly increasing in q (finite scale > 0): the class argmax may compare bytes */
    int nseg;
    int frames;
    float nms_thresh;
    void *dets;            /* device [frames][1000] records of 24 bytes */
    int *counts;           /* device [frames] kept */
    int *raw_counts;       /* device [frames] candidates before NMS (or NULL) */
    int do_nms;
} mhip_detect_t;
int mhip_detect(const mhip_detect_t *p);
int mhip_nms_only(void *dets_dev, int *count_dev, int n, float thresh);

/* ---- image front-end (preproc.hip): letterbox resize + (px - 128); tables from csrc/host/mars_preproc.c */
typedef struct {
    const uint8_t *rgb; size_t rgb_stride;   /* [frames][h][w][3] uint8 on the device */
    int8_t *out;        size_t out_stride;   /* [frames] x (tw*th*3) int8: [th][tw][3] (nhwc) or [3][th][tw] */
    int frames, w, h, tw, th, nhwc;
    int nw, nh, px, py;                      /* resized size and its offset inside the target */
    const int *xstart, *xsrc; const float *xw; /* gather lists of the horizontal / vertical pass (device) */
    const int *ystart, *ysrc; const float *yw;
    int max_cols, max_rows;                  /* largest source region (columns, rows) any 16 x 16 output tile touches; 0 = unknown */
    int n_xtaps;                             /* entries of the horizontal gather list (xstart[nw]); 0 = unknown: the strip form is not offered */
    int max_ytaps;                           /* longest vertical gather list of one output row (the strip form stages 8 of them in LDS); 0 = unknown */
    int form;                                /* 0: the launcher's choice (strips where they fit, else 16 x 16 tiles, else one thread per pixel); 1 / 2: tiles / per-pixel forced (tests) */
} mhip_letterbox_t;
int mhip_letterbox(const mhip_letterbox_t *p);

#ifdef __cplusplus
}
#endif
#endif
