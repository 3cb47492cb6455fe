/*
 * mars_hip.h -- additive extensions of the MI355X build.  Nothing in here
 * exists in the reference; nothing in here changes what the reference API
 * (nna.h, nna_memory.h, nna_tensor.h, mars_runtime.h, mxu_ops.h) does for a
 * single-frame caller.  Plain C ABI: pointers and sizes only.
 *
 * Why they exist:
 *  - the reference has no batch dimension (every kernel ignores shape[0],
 *    reference mars_runtime.c:566-589): mars_hip_set_batch() turns the model's
 *    I/O tensors into `n` independent frames, each computed exactly as one
 *    reference mars_run would compute it;
 *  - the detection tail (decode + NMS) lives in a demo program in the reference
 *    (src/mars/mars_yolo_test.c:80-130), not in its library;
 *  - HBM-resident I/O and one-shot parameter broadcast for one-process-per-GPU
 *    frame sharding.
 */
#ifndef MARS_HIP_H
#define MARS_HIP_H

#include "mars_runtime.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- batching */
/* Re-plans the model for `n` frames.  After the call mars_get_input()->vaddr
 * is a pinned host buffer of n * frame_bytes (frame-major, frames densely
 * packed; frame_bytes = mars_hip_tensor_frame_bytes), ->alloc_size says so; same for outputs.
 * Activations are re-zeroed.  With n == 1 (also the state after mars_load_*) ->alloc_size is what the reference
 * reports: the size of its shared working buffers, i.e. the largest 64-byte-rounded tensor_byte_size() of any
 * activation (reference mars_runtime.c:250-334), and the staging buffer is that large. */
mars_error_t mars_hip_set_batch(mars_model_t *model, int n);
int mars_hip_get_batch(const mars_model_t *model);

/* Bytes of one frame of a tensor by its shape and dtype (numel * element size). */
size_t mars_hip_tensor_frame_bytes(const mars_model_t *model, int tensor_index);
/* The reference's format-aware tensor size (static tensor_byte_size(), mars_runtime.c:80-124): NDHWC32 rounds
 * channels up to 32, NMHWSOIB2 counts 1024-byte blocks, UINT4 packs two per byte.  Host-only. */
size_t mars_hip_tensor_byte_size(const mars_tensor_t *desc);

/* mars_run() = upload + run_device + download.  The three parts: */
mars_error_t mars_hip_upload_inputs(mars_model_t *model);    /* pinned host -> HBM, waits */
mars_error_t mars_hip_run_device(mars_model_t *model);       /* enqueue all layers, waits */
mars_error_t mars_hip_run_device_async(mars_model_t *model); /* enqueue only */
mars_error_t mars_hip_download_outputs(mars_model_t *model); /* HBM -> pinned host, waits */
mars_error_t mars_hip_sync(void);
/* What mars_run() brings back.  MARS_HIP_OUTPUT_HEADS (default): the reference's behaviour, every graph output is copied to
 * mars_get_output()->vaddr (for the 640x640 detectors 2.1 MB per frame: at large batches that copy is what bounds
 * mars_run).  MARS_HIP_OUTPUT_ON_DEVICE: the outputs stay in HBM -- for callers that only want the detections
 * (mars_hip_detect reads the device tensors) or fetch single tensors themselves (mars_hip_read_tensor /
 * mars_hip_download_outputs); vaddr then keeps whatever an earlier download left there. */
#define MARS_HIP_OUTPUT_HEADS 0
#define MARS_HIP_OUTPUT_ON_DEVICE 1
mars_error_t mars_hip_set_output_mode(mars_model_t *model, int mode);

/* Device address / per-frame stride of any tensor (weights: stride 0). */
void *mars_hip_tensor_device(mars_model_t *model, int tensor_index, size_t *frame_stride);
/* Device layout of a tensor's pixel rows: 0 = dense (the reference's bytes).  Otherwise the tensor is a graph output
 * whose [pixels][*row_bytes] rows sit at the returned pitch in HBM (the 255-channel YOLO heads are kept at a 256-byte
 * pitch so that their convolutions store aligned rows); mars_get_output / mars_hip_read_tensor / the detection tail
 * already account for it, only code that reads mars_hip_tensor_device() pointers itself has to. */
int mars_hip_tensor_row_pitch(mars_model_t *model, int tensor_index, int *row_bytes);
/* Copy `bytes` of frame `frame` of any tensor to host memory (debug/parity). */
mars_error_t mars_hip_read_tensor(mars_model_t *model, int tensor_index, int frame, void *dst,
                                  size_t bytes);
/* Write one frame of an activation tensor from host memory (tests). */
mars_error_t mars_hip_write_tensor(mars_model_t *model, int tensor_index, int frame,
                                   const void *src, size_t bytes);

/* Fusion level: 0 = one kernel per reference layer, every tensor materialised
 * (per-layer parity); 1 (default) = conv epilogue fusion of the
 * conv->sigmoid->mul chain and ReLU, copies elided where bit-identical;
 * 2 = 1 + the C3 bottleneck's 1x1 evaluated inside the following 3x3's
 * launch (fewer launches: single-frame latency; no gain at large batches).
 * Must be set before mars_hip_set_batch / first run; re-plans. */
mars_error_t mars_hip_set_fusion(mars_model_t *model, int level);

/* Launch-policy knob of the convolution kernels (process-wide): "persist" (0|1), "persist_stages"
 * (2|3), "persist_maxk", "persist_slots" (0 = what the device holds at once), "stages", "bpx", "variant"
 * (force one launch variant wherever a layer has it); "f32_mfma" (float32 convolutions: 0 = the reference's summation
 * order everywhere, bit-identical; 1 = default: the f32 matrix cores -- fused rounding per tap, inside the 1e-4
 * tolerance of the float32 models -- for every convolution from which no byte-wise MAXPOOL over float bytes is
 * reachable; 2 = those matrix cores everywhere; 3 / 4 = everywhere on the bf16 matrix cores with every operand split,
 * exactly, into two / three bf16 pieces and three / six piece products per product -- same tolerance class, 2.7x the rate:
 * what bench.py --dtype f32 runs is 3); "rows" (default 1: the patch-staged deep-K kernel conv_i8_rows where it is ahead);
 * "few_wgs" (default 256: launches with fewer workgroups than this take the small-launch variant); "graph_max_batch" (default 8: at batches up to this the plan is captured into a
 * HIP graph after its first run and replayed with one call -- single frames are launch-bound; 0 = never);
 * "small_batch" (default 1: a launch whose large-batch tiling yields fewer workgroups than the device has CUs takes
 * smaller tiles -- what single frames want; 0 = the large-batch policy everywhere); "rgb_direct" (default 1: the RGB stem
 * loads its matrix-core operands straight from the image; 0 = the patch-staged form);
 * "dual_stream_min_batch" (default 64: a batch of at least this many frames is enqueued as two halves on two streams,
 * frames being independent, so that the gaps of one half's kernels are filled by the other's; 0 = never);
 * "dual_stream_ways" (2..4 parts, default 2: more parts measured slower).
 * "run_chunk" (default 128: mars_run on a batch of at least twice this many frames goes through in chunks of this size --
 * copy-in of chunk k+1, graph of chunk k and copy-out of chunk k-1 overlap; 0 = one piece).
 * The defaults are the measured optimum; tests use "persist_slots" to force the multi-tile walk
 * of the persistent kernel on small inputs.  Results never depend on these.  0 = ok, -1 = unknown key. */
int mars_hip_set_tuning(const char *key, int value);
int mars_hip_get_tuning(const char *key, int *value); /* the process-wide value in force; -1 = unknown key */
/* The same knobs per model: an override is kept on `model`, put in force for the duration of each of ITS runs (mars_run,
 * mars_hip_run_device[_async], the pipelined submit, mars_hip_autotune) and taken back afterwards, so models that want
 * different launch policies can share a process; mars_hip_set_tuning stays the process default.  At most 16 keys per
 * model.  mars_hip_model_get_tuning returns the override, or the process default where there is none.
 * THREADS: the runtime, like the reference's, is single-threaded: the overrides are put in force by rewriting the process
 * knobs around a run, so runs of different models must not overlap in time on different threads (one thread driving
 * several models, or several threads under one lock, is fine). */
int mars_hip_model_set_tuning(mars_model_t *model, const char *key, int value);
int mars_hip_model_get_tuning(mars_model_t *model, const char *key, int *value);
/* Times the launch variants of every int8 convolution of `model` on the device at the current batch
 * (`reps` launches each, <= 0: 3) and pins the fastest per layer until the plan is rebuilt
 * (set_fusion); call after mars_hip_set_batch.  A one-time load cost; outputs are unaffected. */
mars_error_t mars_hip_autotune(mars_model_t *model, int reps);

/* ------------------------------------------------------- per-layer timing */
/* on = 1: every kernel launch is timed with HIP events on the library's stream (one event per launch: a launch's stop
 * event is the next one's start); on = 2: one event per run of consecutive launches of the same kind -- the run's
 * total is reported on its last launch, the others report 0 (a few events per graph instead of one per launch: what
 * bench.py uses inside its timed region).  Read back after a run. */
void mars_hip_set_profiling(mars_model_t *model, int on);
int mars_hip_num_ops(const mars_model_t *model);
/* kind: 0 conv_i8, 1 conv_f32, 2 elementwise, 3 data movement, 4 other */
int mars_hip_op_info(const mars_model_t *model, int op, int *layer, int *kind, double *macs,
                     double *bytes, float *last_ms);
void *mars_hip_stream(void); /* hipStream_t of the library, as void* */

/* ------------------------------------------------------ parameter arena */
/* One contiguous HBM block holding everything derived from the weight blob
 * (blob mirror, packed conv weights, biases, LUTs).  Identical layout on every
 * rank that loaded the same descriptors, so rank 0 can broadcast it. */
#define MARS_HIP_LOAD_DEFER_WEIGHTS 1u /* parse descriptors, leave the arena unfilled */
/* The launch plan of a file as text, computed on the host only (no device needed: planner tests on the CPU, debugging): one line per launch
 * ("op I layer L KIND in T.. out T" + its flags: relayout / planar_store / lut / add / seg / pair_next / k_limit / in_rec / out_rec / rows_only ...)
 * and one per tensor that is not held as tagged on the device (nhwc_c / partial / zero_from / rec_c / pix_stride).  The tuning and environment in force
 * decide as for a load.  Returns the length of the text (cut at cap, always terminated); 0 = the loader rejects the file. */
size_t mars_hip_describe_plan(const void *data, size_t size, unsigned flags, char *out, size_t cap);
mars_error_t mars_hip_load_memory_ex(const void *data, size_t size, unsigned flags,
                                     mars_model_t **model);
void *mars_hip_param_arena(mars_model_t *model, size_t *bytes);

/* ------------------------------------------------------- detection tail */
/* Same record as det_t of reference mars_yolo_test.c:37 (24 bytes). */
typedef struct {
    float x, y, w, h, conf;
    int cls;
} mars_det_t;

#define MARS_YOLO_MAX_DET 1000 /* candidate cap of the reference demo (:187-189) */

/* Host-pointer forms with the reference's exact semantics (one frame):
 * parse_output (:80-104) and nms (:107-130), executed on the GPU. */
int mars_yolo_parse_output(const int8_t *data, int npred, float scale, mars_det_t *dets, int maxd);
int mars_yolo_nms(mars_det_t *dets, int n, float thresh);

/* Batched, device-resident: decode + NMS over the model's current batch.
 * The prediction list of a frame is the concatenation, in the order given, of
 * the listed output tensors, each viewed as rows of 85 int8 with its own
 * desc.scale.  dets: host buffer [batch][MARS_YOLO_MAX_DET]; counts: [batch]
 * (kept detections, sorted as the reference leaves them). */
mars_error_t mars_hip_detect(mars_model_t *model, const int *output_indices, int n_outputs,
                             float nms_thresh, mars_det_t *dets, int *counts);
/* Same, but results stay in HBM (no D2H) -- used by the benchmark loop. */
mars_error_t mars_hip_detect_device(mars_model_t *model, const int *output_indices, int n_outputs,
                                    float nms_thresh);

/* ------------------------------------------------------- pipelined host I/O */
/* mars_run() pays host->HBM copy, graph and HBM->host copy one after the other.  With an open pipe up to three batches are
 * in flight: batch k+1 uploads and batch k-1 downloads (on their own streams, from / into their own buffers) while batch k
 * is computed.  Usage (batch size = mars_hip_set_batch, unchanged while the pipe is open):
 *     mars_hip_pipe_open(model, &opts);
 *     for every batch:  fill mars_hip_pipe_input(model, 0)  ->  mars_hip_pipe_submit(model)
 *                       [from the third batch on]  mars_hip_pipe_wait(model, outs, &dets, &counts)  -> use results
 *     mars_hip_pipe_wait() once more per batch still in flight; mars_hip_pipe_close(model).
 * Results of a batch are bit for bit what mars_run() / mars_hip_detect() give for the same input. */
typedef struct {
    int download_outputs;   /* 1: the graph outputs come back (dense [batch][frame bytes] each, as mars_run leaves them) */
    int detect;             /* 1: decode + NMS on the device, detections come back ([batch][MARS_YOLO_MAX_DET] + counts) */
    int det_outputs[4];     /* detect: output indices forming the prediction list, as for mars_hip_detect */
    int n_det_outputs;
    float nms_thresh;
    int camera_w, camera_h; /* both > 0: CAMERA mode.  Graph input 0 (int8, 3 channels) is fed from uint8 RGB frames of camera_w x
                             * camera_h: mars_hip_pipe_input(model, 0) is then a pinned buffer of [batch][camera_h][camera_w][3] bytes,
                             * and a submit uploads it and runs the letterbox / px - 128 front-end (the reference's load_image(),
                             * src/mars/mars_yolo_test.c:40-77, as mars_hip_preprocess does it) on the main stream ahead of the graph
                             * (round 6: on the upload stream it held up the next batch's copy), into the
                             * slot's graph input -- the demo's whole loop (:132-214: load_image -> run -> parse_output -> nms),
                             * three batches in flight.  0 / 0: the graph inputs themselves are uploaded (as before).
                             * The struct has grown by these two fields (round 5): callers zero-initialise it (`= {0}` / memset) before
                             * setting what they use, so that a build against an older header still means "no camera" (ADVICE r5). */
} mars_hip_pipe_opts_t;
mars_error_t mars_hip_pipe_open(mars_model_t *model, const mars_hip_pipe_opts_t *opts);
/* pinned host buffer ([batch][frame bytes]) to fill for the NEXT submit; changes after every submit */
void *mars_hip_pipe_input(mars_model_t *model, int input_index);
/* queue upload + graph (+ tail) + download of the buffer just filled; returns at once.  At most three batches may be
 * in flight: MARS_ERR_ALLOC_FAILED asks for a mars_hip_pipe_wait first */
mars_error_t mars_hip_pipe_submit(mars_model_t *model);
/* block until the OLDEST submitted batch is complete.  outputs (may be NULL): array of mars_get_num_outputs() pointers,
 * set to that batch's host copies (download_outputs); dets / counts (may be NULL): its detections (detect).  The
 * buffers stay valid until the SECOND mars_hip_pipe_submit after this call returns (four buffer sets, three batches in
 * flight): in the steady-state loop "wait k; submit k+3; use results of k" the results are safe while k+3 is queued and
 * are overwritten by the submit after that. */
mars_error_t mars_hip_pipe_wait(mars_model_t *model, const void **outputs, const mars_det_t **dets, const int **counts);
void mars_hip_pipe_close(mars_model_t *model);
/* camera mode: device time (ms) of the image front-end kernel of the batch mars_hip_pipe_wait() handed out last; < 0 = not available */
float mars_hip_pipe_camera_ms(mars_model_t *model);

/* --------------------------------------------------- synthetic .mars writer */
/* Well-formed graphs (NHWC activations, OHWI int8 weights, int32 bias; or
 * NCHW/OIHW float32) with the YOLOv5 layer sequence and seeded weights, for
 * the model files the reference repo does not ship (.MISSING_LARGE_BLOBS).
 * Format restated from reference include/mars.h:103-221. */
typedef struct {
    int width_x16;   /* channel multiple: 4 = yolov5n (0.25), 8 = yolov5s (0.50) */
    int depth_x3;    /* depth multiple in thirds: 1 = n/s (0.33) */
    int input_hw;    /* square input, multiple of 32 */
    int float32;     /* 0: int8 NHWC; 1: float32 NCHW */
    int nchw_int8;   /* int8 only: emit NCHW/OIHW tags instead of NHWC/OHWI */
    unsigned seed;
    int tiny;        /* 1: 3-conv "tiny_160" chain instead of YOLOv5 */
    int vary_scales; /* 1: every convolution gets its own output / sigmoid / SiLU scales (x0.75 .. x1.35 of the nominal
                      * ones), so no two fused tables are equal; 0 keeps the files of earlier versions byte for byte */
} mars_synth_opts_t;
/* returns the file size; writes at most cap bytes (call with cap 0 to size) */
size_t mars_synth_model(const mars_synth_opts_t *opts, void *buf, size_t cap);

/* ------------------------------------------------------- image front-end */
/* The reference's load_image() (src/mars/mars_yolo_test.c:40-77) after the file decode, on the GPU: letterbox
 * resize of a uint8 RGB image [h][w][3] to tw x th exactly as stbir_resize_uint8() of the vendored
 * stb_image_resize.h does it (Catmull-Rom up / Mitchell down, clamped edges, linear), grey (-17) padding,
 * px - 128.  out = int8 [th][tw][3] (nhwc != 0) or [3][th][tw].  Host pointers; 0 = ok, -1 = failure. */
int mars_yolo_letterbox(const unsigned char *rgb, int w, int h, int tw, int th, int nhwc, signed char *out);
/* Camera batch: `frames` RGB frames of w x h (host memory, contiguous) are letterboxed straight into frames
 * [first_frame, first_frame + frames) of graph input `input_index` in HBM, in the layout its format tag asks for
 * (mars_yolo_test.c:157-165); follow with mars_hip_run_device().  Input must be int8 with 3 channels. */
mars_error_t mars_hip_preprocess(mars_model_t *model, int input_index, const unsigned char *rgb_frames, int w, int h,
                                 int first_frame, int frames);
/* The same front-end on frames that are ALREADY in device memory (a capture card writing into HBM, or a caller's own upload):
 * rgb_dev = frames x [h][w][3] bytes, contiguous; enqueued on the library's current stream, no synchronisation. */
mars_error_t mars_hip_preprocess_device(mars_model_t *model, int input_index, const void *rgb_dev, int w, int h,
                                        int first_frame, int frames);

#ifdef __cplusplus
}
#endif
#endif /* MARS_HIP_H */
