/*
 * mars_internal.h -- private state of the MI355X .mars executor (host side).
 * The public prefix of mars_model_ext_t is the reference's mars_model_t
 * (include/mars_runtime.h), so a mars_model_t* handed to callers is also a
 * pointer to the private record.
 */
#ifndef MARS_INTERNAL_H
#define MARS_INTERNAL_H

#include <stddef.h>
#include <stdint.h>

#include "../mhip.h"
#include "mars_hip.h"
#include "mars_runtime.h"

enum {
    OP_CONV_I8 = 0,
    OP_CONV_F32,
    OP_RELU_BYTES,
    OP_LUT_I8,
    OP_BINARY_I8,
    OP_SIGMOID_F32,
    OP_BINARY_F32,
    OP_RELU_F32,
    OP_BN,
    OP_MAXPOOL,
    OP_CONCAT_SLICE,
    OP_UPSAMPLE,
    OP_UPSAMPLE_Q, OP_MAXPOOL_Q, /* UPSAMPLE / stride-1 MAXPOOL of an NCHW-tagged graph on tensors held pixels x channels (nhwc_internal) */
    OP_CONCAT_Q, /* a whole CONCAT layer of an NCHW-tagged graph on tensors held pixels x channels (nhwc_internal; mhip_concat_nchwq) */
    OP_CONV_F32_VHEAD, /* the first pixels of a conv_f32 that reads a never-materialised float CONCAT (virtual_concat_f32; mhip_conv_f32_vcat_head) */
    OP_FAIL, /* mars_run stops here with op->err, as the reference would at this layer */
};

#define NO_OFF ((size_t)-1)
#define MARS_MAX_IO 4 /* graph inputs / outputs (header arrays of the .mars format) */
#define MARS_MAX_MODEL_TUNE 16 /* per-model tuning overrides (mars_hip_model_set_tuning) */

typedef struct {
    size_t bytes;    /* numel * elemsize by shape */
    size_t extent;   /* largest byte offset + 1 any op touches (>= bytes) */
    size_t stride;   /* device bytes per frame (0 for weights) */
    uint8_t *dev;    /* device base */
    int is_weight;
    int needed;      /* touched by an op, or graph input/output */
    int io_in, io_out; /* 1-based graph input / output slot, 0 if none */
    uint8_t *host;   /* pinned staging for graph I/O: batch * bytes */
    uint8_t *dense_dev; /* padded rows only: batch * bytes, the dense copy made on the device before a download */
    int pix_c, pix_stride; /* pix_stride != 0: [pixels][pix_c] rows kept at a pix_stride-byte pitch on the device (pad_output_rows) */
    int rec_c, rec_hw;     /* rec_c != 0: a float tensor kept in RECORD format on the device (rec_pairs): [rec_c / 8][rec_hw] x 32 bytes */
    int nhwc_c, nhwc_hw;   /* nhwc_c != 0: an NCHW-tagged int8 tensor kept as [nhwc_hw][nhwc_c] (pixels x channels) on the device (nhwc_internal) */
    size_t zero_from;      /* != 0: bytes from this offset on are never written by any layer (zero_tail_f32 relies on it): mars_hip_write_tensor refuses non-zero bytes there */
    int partial;           /* only part of the tensor is ever written (virtual_concat_q keeps the first rows of a concat): not readable through mars_hip_read_tensor */
    int nhwc_pitch;        /* ... bytes between its pixels (0 = nhwc_c; a write-only 255-channel head is kept at 256) */
} mtensor_t;

typedef struct {
    int kind, layer, err;
    int t_in[4], n_in, t_out; /* tensor indices, -1 if none */
    /* geometry (conv / pool / concat / upsample share these) */
    int in_h, in_w, in_c, out_h, out_w, out_c, kh, kw, sh, sw, pt, pl;
    int nchw, relu, is_mul, is_f32, leaky, safe; /* nchw (conv_i8): the INPUT is [C][H][W] bytes and is relaid into scratch before the launch */
    size_t in_byte_off, out_byte_off; /* conv_i8: added to the input / output tensor's address (virtual_concat_q: a convolution over a row range of a tensor) */
    int k_limit;   /* conv_f32 (1 x 1): input channels >= k_limit are exact zeros in every frame (zero_tail_f32): mhip_conv_f32_t.k_limit under modes 3 / 4; 0 = none */
    int vc_n, vc_t[3], vc_run, vc_shift; /* conv_f32 (1 x 1) over a float CONCAT that is never materialised (virtual_concat_f32): t_in[0] is the concat's LAST input, read
                      through a view vc_shift bytes in front of it, K loop = k_limit planes (required);  OP_CONV_F32_VHEAD: the concat's other inputs
                      (vc_n of them, runs of vc_run floats) and k_limit = planes to sum for the first vc_n * vc_run pixels */
    int rows_only; /* OP_CONCAT_Q: only the first `rows_only` map rows of the output are produced (the rest of it is never read: virtual_concat_q); 0 = all */
    int out_nchw;  /* conv_i8: the result is stored [O][H][W] (the reference's conv2d_int8_mxu); both set by the input's tag, cleared per side by nhwc_internal */
    int silu_f32;  /* conv_f32 with the float SIGMOID + MUL pair (ONNX SiLU) folded into its epilogue */
    int f32_exact; /* conv_f32 whose result reaches a byte-wise MAXPOOL over float bytes: keeps the reference's summation order */
    int variant; /* conv_i8 launch variant pinned by mars_hip_autotune (0 = default policy) */
    int add_t; float add_s_conv, add_s_other, add_inv; /* conv_i8 with a residual Add folded in: other operand (tensor index + 1, 0 = none) */
    int nseg, seg_t[4], seg_c[4], seg_up;
    int chain_n, chain_out[3]; /* OP_MAXPOOL heading a fused chain of stride-1 pools: every stage's output tensor */
    int pair_next; /* conv_i8: launched together with the NEXT op (same input, same geometry: C3's cv1 + cv2) */ /* conv_i8 reading a never-materialised concat: its segments (tensor, channels) */
    int row_pad, oc_pad, c_pad;
    int ch_off, scale_h, scale_w, bn_n;
    int out_pix_stride, out_ch_off; /* producer writes a channel slice of a wider tensor (zero-copy concat) */
    int store_c; /* conv_i8 writing padded pixel rows: channels stored per pixel (the padded count), 0 = out_c */
    float cs, f0, f1, f2;
    size_t n;                 /* elements per frame for element-wise ops */
    size_t w_off, b_off, lut_off, s_off; /* offsets into the parameter arena */
    int pre;         /* fused bottleneck: a 1x1 + SiLU evaluated on the staged patch before this convolution (fuse_bottleneck) */
    size_t pre_w_off, pre_b_off, pre_lut2_off;
    float pre_cs;
    size_t w2_off;   /* a second image of the weights, or NO_OFF: the RGB stem's as conv_i8_rgb keeps them in LDS
                        (mhip_conv_i8_rgb_pack), or (w2_rows) a deep 3x3 layer's as conv_i8_rows streams them (mhip_conv_i8_rows_pack) */
    int w2_rows;
    int w2_planes;   /* conv_f32: bf16 planes packed into the w2 image at load (2: hi, mid -- f32_mfma 3; 3: + lo -- f32_mfma 4); 0 = none */
    size_t w3_off;   /* conv_f32: conv_f32_patch's image (unit table, schedule, two bf16 planes in its K order), or NO_OFF */
    int w3_stem;     /* ... that image is conv_f32_stem's */
    int in_rec, out_rec; /* conv_f32: the input / output tensor is in record format (rec_pairs; mhip_conv_f32_t.in_rec / out_rec) */
    size_t lut2_off; /* 512-entry half-step form of the fused LUT (4-instruction requantisation), or NO_OFF */
    size_t w_blob_off[2];     /* operands that live in the blob mirror */
    double macs, bytes;       /* algorithmic work per frame */
    int prof_kind;
    float last_ms;
    void *ev0, *ev1;
    int prof_rec;   /* profiling: this launch's stop event was recorded in the last run */
    void *ev_start; /* profiling: the event that marks this launch's start = the previous launch's ev1 (own ev0 for the first) */
} mars_op_t;

typedef struct mars_model_ext {
    mars_model_t pub; /* MUST stay first */
    int batch, fusion, profiling, deferred;
    int plan_err;   /* first allocation failure while planning (build_plan returns it; 0 = none) */
    int no_vconcat; /* virtual concat switched off (a batch too large for 32-bit buffer offsets) */
    int no_download; /* mars_hip_set_output_mode(MARS_HIP_OUTPUT_ON_DEVICE): mars_run leaves the graph outputs in HBM */
    int no_bottleneck; /* fused bottlenecks (fusion level 2) switched off: one of them cannot launch at this batch */
    int rec_frames;    /* rec_pairs: the batch the record-format pairs were chosen for (0 = one frame); a pair whose tensors reach 4 GiB at that batch is left alone */
    int rec_skipped;   /* ... some pair was left alone for that reason (a smaller batch may take it) */
    size_t rec_max_frames; /* ... the largest batch every chosen pair still fits */
    mtensor_t *mt;
    mars_op_t *ops;
    int n_ops, cap_ops;
    /* parameter arena: host image + device copy */
    uint8_t *arena_host;
    size_t arena_size, arena_cap;
    uint8_t *arena_dev;
    size_t blob_mirror_bytes;
    /* activations */
    uint8_t *act_dev;
    size_t act_bytes;
    uint8_t *scratch_dev;
    size_t scratch_per_frame;
    int scratch_t, scratch_cpad, scratch_f0, scratch_n; /* what the relayout scratch holds: tensor (-1 = nothing), channel padding, frame range (enqueue_range) */
    /* detection tail */
    void *det_dev;
    int *det_counts_dev;
    float *det_lut_dev;
    int det_cap;
    float det_lut_scale[4]; /* scales the uploaded decode LUTs were built for */
    int det_lut_n;
    int det_lut_mono[4];    /* value table of that segment strictly increasing (mhip_detect_t.mono) */
    void *ev_graph_done, *ev_tail_done; /* main->aux and aux->main hand-offs */
    int tail_pending;
    int frame0, run_frames; /* frame range the launches being enqueued cover (a large batch runs as two halves on two streams) */
    void *ev_fork, *ev_join[3];
    void *ev_chunk[2][8]; /* mars_run at large batches: [0] chunk uploaded, [1] chunk computed */
    void *graph_exec;   /* captured HIP graph of the plan at the current batch (small batches), or NULL */
    unsigned graph_gen; /* tuning generation it was captured under */
    int ran_plain;      /* the plan has run launch by launch at this batch: every launcher's one-time set-up is done */
    void *pipe; /* double-buffered I/O state (mars_pipe.c), NULL when closed */
    struct { char key[28]; int value, saved; } tune[MARS_MAX_MODEL_TUNE]; /* launch-policy overrides of this model */
    int n_tune, tune_depth;
    int plan_f32_mode; /* the f32_mfma mode build_plan ran under (weight images, record pairs): replan_for_f32_mode */
    struct mars_model_ext *live_next; /* every loaded model, newest first (mars_live_models): a process-wide mode change re-plans the float ones */
} mars_model_ext_t;


/* ---- shared between mars_model.c (loader), mars_plan.c (planner) and mars_run.c (run paths); hidden from the library's ABI */
#define MARS_INTERNAL __attribute__((visibility("hidden")))
#define ALIGN_UP(x, a) (((x) + (size_t)(a) - 1) & ~((size_t)(a) - 1))
#define NO_TENSOR 0xFFFFFFFFu
#define MAX_DIM_PRODUCT ((size_t)1 << 40)
#define ARENA_MAX ((size_t)1 << 40) /* parameter arena: anything beyond is a corrupt file, not a model */
#define MAX_CHANNELS 65536    /* per-tensor channel count the planners accept */
MARS_INTERNAL int mars_verbose(void);
#define VLOG(...) do { if (mars_verbose()) fprintf(stderr, "Mars: " __VA_ARGS__); } while (0)
/* mars_model.c */
MARS_INTERNAL void drop_graph(mars_model_ext_t *m);
MARS_INTERNAL mars_model_ext_t *mars_live_models(void);
MARS_INTERNAL mars_error_t build_plan(mars_model_ext_t *m);
MARS_INTERNAL mars_error_t upload_params(mars_model_ext_t *m);
MARS_INTERNAL mars_error_t alloc_batch(mars_model_ext_t *m, int n);
/* mars_plan.c */
MARS_INTERNAL size_t elem_size(uint32_t dtype);
MARS_INTERNAL size_t shape_numel(const mars_tensor_t *d);
MARS_INTERNAL size_t reference_buffer_size(const mars_model_ext_t *m);
MARS_INTERNAL size_t arena_reserve(mars_model_ext_t *m, size_t bytes);
MARS_INTERNAL void blob_read(const mars_model_ext_t *m, size_t off, size_t n, void *dst);
MARS_INTERNAL void plan_layer(mars_model_ext_t *m, int li);
MARS_INTERNAL void nhwc_internal(mars_model_ext_t *m);
MARS_INTERNAL void virtual_concat_q(mars_model_ext_t *m);
MARS_INTERNAL void zero_tail_f32(mars_model_ext_t *m);
MARS_INTERNAL void virtual_concat_f32(mars_model_ext_t *m);
MARS_INTERNAL void fuse_silu(mars_model_ext_t *m);
MARS_INTERNAL void fuse_lut(mars_model_ext_t *m);
MARS_INTERNAL void fuse_silu_f32(mars_model_ext_t *m);
MARS_INTERNAL void elide_concat(mars_model_ext_t *m);
MARS_INTERNAL void fuse_add(mars_model_ext_t *m);
MARS_INTERNAL void fuse_add_f32(mars_model_ext_t *m);
MARS_INTERNAL void rec_pairs(mars_model_ext_t *m);
MARS_INTERNAL void pair_convs_f32(mars_model_ext_t *m);
MARS_INTERNAL void trim_concat(mars_model_ext_t *m);
MARS_INTERNAL int mars_preproc_prepare(int w, int h, int tw, int th); /* mars_preproc.c: gather tables of a letterbox geometry, cached */
MARS_INTERNAL void fuse_bottleneck(mars_model_ext_t *m);
MARS_INTERNAL void virtual_concat(mars_model_ext_t *m);
MARS_INTERNAL void pair_convs(mars_model_ext_t *m);
MARS_INTERNAL void fuse_pool_chains(mars_model_ext_t *m);
MARS_INTERNAL void f32_policy(mars_model_ext_t *m);
MARS_INTERNAL void rec_pairs(mars_model_ext_t *m);
/* mars_run.c */
MARS_INTERNAL void conv_i8_params(const mars_model_ext_t *m, const mars_op_t *op, mhip_conv_i8_t *p);

/* detection tail pieces shared with the pipelined I/O (mars_yolo.c) */
mars_error_t mars_detect_prepare(mars_model_ext_t *m, const int *output_indices, int n_outputs);
int mars_detect_launch(mars_model_ext_t *m, const int *output_indices, int n_outputs, float nms_thresh, void *dets_dev,
                       int *counts_dev);

/* shared host helpers (mars_model.c) */
int32_t mars_trunc_x86(float x);
void mars_pack_conv_i8(const int8_t *w, size_t avail, int nchw, int out_c, int in_c, int kh, int kw,
                       int c_pad, int row_pad, int oc_pad, int8_t *dst);

#endif
