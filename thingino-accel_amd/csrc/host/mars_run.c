/*
 * mars_run.c -- the run paths of the .mars executor: per-op launch (plan entry -> C-ABI launcher of csrc/mhip.h), stream
 * scheduling (two half-batches on two streams, HIP-graph replay of small batches), mars_run / mars_hip_run_device, host I/O,
 * tuning knobs (process-wide and per model), the autotuner, tensor access and profiling.
 * Reference: src/mars/mars_runtime.c:439-459 (mars_run) and the dispatcher :1161-1224.
 * Split out of mars_model.c in round 4; loader: mars_model.c, planner: mars_plan.c.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mars_internal.h"
#include "nna.h"

/* single frames / small batches are launch-bound (60 launches of a few microseconds): their plan is captured into a HIP
 * graph after the first plain run and replayed.  g_graph_max_batch: largest batch that takes this path (0 = off);
 * g_tune_gen: bumped by every tuning call, so that graphs captured under older launch policies are dropped. */
static int g_graph_max_batch = 8;
static unsigned g_tune_gen = 1;
/* Batches of at least this many frames run as TWO halves on two streams (0 = never): frames are independent, and two
 * graph instances in different layers fill each other's gaps -- waves parked at barriers / DMA waits (45 % of wave time
 * in every convolution kernel, profiles/r02_mfma_busy.json) and the tail of every launch.  Measured on the yolov5s twin,
 * batch 256: 4.76 -> 4.58 ms per batch. */
static int g_dual_min_batch = 64;
static int g_dual_ways = 2; /* parts (= streams) such a batch is cut into: 2..4 */
/* (Round 3, measured and dropped: a DEPTH-FIRST head -- the first 2 / 3 / 5 / 8 launches of the plan run in chunks of
 * 16 / 32 / 64 frames, so that the stem's 3.3 MB per frame is still in the 256 MB Infinity Cache when the next layer reads
 * it: 4.55-4.65 ms per batch against 4.54-4.59 without, at every setting.  The early layers are not waiting for HBM reads.) */
/* (Round 3, measured and dropped, twice: a LAZY join -- the main stream not waiting for the other part at the end of a run,
 * only the detection tail, uploads and downloads doing so -- so that back-to-back runs keep both streams busy without
 * meeting at every run boundary: 1-2 % slower at batch 256 and 128.  And the same with a deliberate OFFSET: the second part
 * started once, 8 / 16 / ... / 48 launches behind the first, so that one stream sits in the early HBM-bound layers while the
 * other is in the deep matrix-bound ones from then on: 4.45-4.50 ms per batch at every offset against 4.43 joined, 2.47
 * against 2.37 at batch 128.  The aligned start of the two halves is worth more than the bubble at the join costs.) */
/* ------------------------------------------------------------------ running */
/* device address of the first frame of the range being enqueued (weights: one copy for every frame) */
static uint8_t *tdev(const mars_model_ext_t *m, int ti) {
    if (ti < 0 || !m->mt[ti].dev) return NULL;
    return m->mt[ti].dev + (m->mt[ti].is_weight ? 0 : (size_t)m->frame0 * m->mt[ti].stride);
}
static size_t tstride(const mars_model_ext_t *m, int ti) { return ti >= 0 ? m->mt[ti].stride : 0; }

void conv_i8_params(const mars_model_ext_t *m, const mars_op_t *op, mhip_conv_i8_t *p) {
    uint8_t *A = m->arena_dev;
    memset(p, 0, sizeof(*p));
    p->in = (const int8_t *)tdev(m, op->t_in[0]); p->in_stride = tstride(m, op->t_in[0]);
    p->in_c = op->in_c;
    p->out = (int8_t *)tdev(m, op->t_out); p->out_stride = tstride(m, op->t_out);
    if (p->in) p->in += op->in_byte_off;   /* (a row range of the tensors: virtual_concat_q) */
    if (p->out) p->out += op->out_byte_off;
    p->w = (const int8_t *)(A + op->w_off);
    p->bias = op->b_off != NO_OFF ? (const int32_t *)(A + op->b_off) : NULL;
    p->lut = op->lut_off != NO_OFF ? A + op->lut_off : NULL;
    p->lut2 = op->lut_off != NO_OFF && op->lut2_off != NO_OFF ? A + op->lut2_off : NULL;
    p->w_rgb = op->w2_off != NO_OFF && !op->w2_rows ? (const int8_t *)(A + op->w2_off) : NULL;
    p->w_rows = op->w2_off != NO_OFF && op->w2_rows ? (const int8_t *)(A + op->w2_off) : NULL;
    if (op->pre) {
        p->pre_w = (const int8_t *)(A + op->pre_w_off);
        p->pre_bias = (const int32_t *)(A + op->pre_b_off);
        p->pre_lut2 = A + op->pre_lut2_off;
        p->pre_cs = op->pre_cs;
    }
    p->frames = m->run_frames;
    p->in_h = op->in_h; p->in_w = op->in_w;
    p->out_h = op->out_h; p->out_w = op->out_w; p->out_c = op->store_c ? op->store_c : op->out_c;
    p->kh = op->kh; p->kw = op->kw; p->stride_h = op->sh; p->stride_w = op->sw; p->pad_top = op->pt; p->pad_left = op->pl;
    p->row_pad = op->row_pad; p->oc_pad = op->oc_pad; p->cs = op->cs; p->relu = op->relu; p->out_nchw = op->out_nchw;
    p->safe = op->safe;
    p->out_pix_stride = op->out_pix_stride; p->out_ch_off = op->out_ch_off;
    p->variant = op->variant;
    if (op->add_t && tstride(m, op->add_t - 1) == p->out_stride) {
        p->add = (const int8_t *)tdev(m, op->add_t - 1);
        p->add_s_conv = op->add_s_conv; p->add_s_other = op->add_s_other; p->add_inv = op->add_inv;
    }
    p->nseg = op->nseg;
    p->seg_up = op->seg_up;
    int c0 = 0;
    for (int k = 0; k < 4; k++) {
        p->seg_c0[k] = 0x7fffffff;
        if (k < op->nseg) {
            p->seg_in[k] = (const int8_t *)tdev(m, op->seg_t[k]);
            p->seg_stride[k] = tstride(m, op->seg_t[k]);
            p->seg_c[k] = op->seg_c[k];
            p->seg_c0[k] = c0;
            c0 += op->seg_c[k];
        }
    }
    if (op->nseg > 1) p->in = p->seg_in[0];
}

static void conv_f32_params(mars_model_ext_t *m, mars_op_t *op, mhip_conv_f32_t *p) {
    const int B = m->run_frames;
    uint8_t *A = m->arena_dev;
    memset(p, 0, sizeof(*p));
    p->in = (const float *)tdev(m, op->t_in[0]); p->in_stride = tstride(m, op->t_in[0]);
    p->out = (float *)tdev(m, op->t_out); p->out_stride = tstride(m, op->t_out);
    p->w = (const float *)(A + op->w_off);
    p->w_split = op->w2_off != NO_OFF ? (const void *)(A + op->w2_off) : NULL;
    p->w_patch = op->w3_off != NO_OFF ? (const void *)(A + op->w3_off) : NULL;
    p->bias = op->b_off != NO_OFF ? (const float *)(A + op->b_off) : NULL;
    p->frames = B;
    p->in_h = op->in_h; p->in_w = op->in_w; p->in_c = op->in_c;
    p->out_h = op->out_h; p->out_w = op->out_w; p->out_c = op->out_c;
    p->kh = op->kh; p->kw = op->kw; p->stride_h = op->sh; p->stride_w = op->sw; p->pad_top = op->pt; p->pad_left = op->pl;
    p->silu = op->silu_f32;
    if (op->add_t) { p->add = (const float *)tdev(m, op->add_t - 1); p->add_stride = tstride(m, op->add_t - 1); }
    {
        const int mode = mhip_conv_f32_mode(-1);
        p->use_mfma = mode == 3 ? 3 : mode == 4 ? 2 : (mode == 2 || (mode == 1 && !op->f32_exact));
    }
    if (op->in_rec || op->out_rec) { /* planned under f32_mfma = 3 (rec_pairs): nothing but that mode's kernels reads / writes records.  A mode change re-plans the
                                      * model (replan_for_f32_mode); this only holds for descriptor-only ranks, which cannot */
        p->in_rec = op->in_rec; p->out_rec = op->out_rec;
        p->use_mfma = 3;
    }
    /* the w2 image holds the planes of the mode in force AT LOAD (two under f32_mfma = 3, three under 4).  conv_f32_split must never
     * read a third plane that was not packed (the next op's arena bytes): without the planes the launch goes to the f32 matrix cores
     * (conv_f32_mfma), as for a model loaded under modes 0 - 2.  Likewise conv_f32_patch / conv_f32_stem images are mode 3's. */
    {
        const int need = p->use_mfma == 2 ? 3 : p->use_mfma == 3 ? 2 : 0;
        if (need > op->w2_planes) p->w_split = NULL;
        if (p->use_mfma != 3) p->w_patch = NULL;
    }
    if (p->use_mfma >= 2) p->k_limit = op->k_limit; /* (the exact-order and f32-matrix-core kernels sum every term, as the reference does) */
    if (op->vc_shift && op->kind == OP_CONV_F32) { /* virtual_concat_f32: the concat's last input seen vc_shift bytes early; what lies behind k_limit planes is NOT zero */
        p->in = p->in ? (const float *)((const uint8_t *)p->in - op->vc_shift) : NULL;
        p->k_limit = op->k_limit;
        p->k_limit_required = 1; /* (planned under modes 3 / 4; any other kernel choice fails the launch instead of summing those planes) */
        if (p->use_mfma < 2 || !p->w_split) p->in = NULL;
    }
}

static int launch_op(mars_model_ext_t *m, mars_op_t *op) {
    const int B = m->run_frames;
    uint8_t *A = m->arena_dev;
    switch (op->kind) {
        case OP_CONV_I8: {
            mhip_conv_i8_t p;
            conv_i8_params(m, op, &p);
            if (op->nchw && op->c_pad == 4 && op->in_c <= 4 && !op->add_t) {
                /* the small-channel stem of an NCHW-tagged graph: conv_i8_smallc reads the planes themselves and interleaves them while it stages
                 * its patch (round 6: no relayout launch, 0.13 ms of yolov5n_int8.mars' 2.7 ms at batch 256); -2 = it does not take the shape */
                if (!getenv("MARS_HIP_NO_PLANAR_STEM")) { /* (A / B switch and tests; read per launch: one getenv per run) */
                    mhip_conv_i8_t q = p;
                    q.in_c = op->c_pad;
                    q.in_planar = op->in_c;
                    const int rc = mhip_conv_i8(&q);
                    if (rc != -2) return rc;
                }
            }
            if (op->nchw) {
                const size_t ss = ALIGN_UP(m->scratch_per_frame, 256);
                int8_t *scratch = (int8_t *)m->scratch_dev + (size_t)m->frame0 * ss;
                /* two convolutions in a row over the same tensor (C3's cv1 + cv2) share one relayout: the scratch still holds it when
                 * nothing has written the tensor since (enqueue_range drops the tag at every write and at the start of a range) */
                if (!(m->scratch_t == op->t_in[0] && m->scratch_cpad == op->c_pad && m->scratch_f0 == m->frame0 && m->scratch_n == B)) {
                    int rc = mhip_nchw_to_nhwc_pad(p.in, p.in_stride, scratch, ss, B, op->in_c, op->in_h * op->in_w, op->c_pad);
                    if (rc) return rc;
                    m->scratch_t = op->t_in[0]; m->scratch_cpad = op->c_pad; m->scratch_f0 = m->frame0; m->scratch_n = B;
                }
                p.in = scratch; p.in_stride = ss; p.in_c = op->c_pad;
            }
            if (op->add_t && !p.add) return -1; /* planner guaranteed equal strides */
            return mhip_conv_i8(&p);
        }
        case OP_CONV_F32: {
            mhip_conv_f32_t p;
            conv_f32_params(m, op, &p);
            return mhip_conv_f32(&p);
        }
        case OP_CONV_F32_VHEAD: {
            mhip_conv_f32_t p;
            const float *first[3] = {NULL, NULL, NULL};
            size_t strides[3] = {0, 0, 0};
            conv_f32_params(m, op, &p);
            if (p.use_mfma < 2) return -1; /* (the plan is the split-bf16 modes': replan_for_f32_mode) */
            p.k_limit = op->k_limit;
            for (int k = 0; k < op->vc_n && k < 3; k++) {
                first[k] = (const float *)tdev(m, op->vc_t[k]);
                strides[k] = tstride(m, op->vc_t[k]);
            }
            if (op->w3_off == NO_OFF) return -1;
            return mhip_conv_f32_vcat_head(&p, (const float *)(A + op->w3_off), first, strides, op->vc_n, op->vc_run);
        }
        case OP_RELU_BYTES:
            return mhip_relu_bytes((int8_t *)tdev(m, op->t_out), tstride(m, op->t_out), B, op->n);
        case OP_LUT_I8:
            return mhip_lut_i8((const int8_t *)tdev(m, op->t_in[0]), tstride(m, op->t_in[0]), (int8_t *)tdev(m, op->t_out),
                               tstride(m, op->t_out), B, op->n, A + op->lut_off);
        case OP_BINARY_I8:
            return mhip_binary_i8(op->is_mul, (const int8_t *)tdev(m, op->t_in[0]), tstride(m, op->t_in[0]),
                                  (const int8_t *)tdev(m, op->t_in[1]), tstride(m, op->t_in[1]),
                                  (int8_t *)tdev(m, op->t_out), tstride(m, op->t_out), B, op->n, op->f0, op->f1, op->f2,
                                  op->out_pix_stride ? op->in_c : 0, op->out_pix_stride, op->out_ch_off);
        case OP_SIGMOID_F32:
            return mhip_sigmoid_f32((const float *)tdev(m, op->t_in[0]), tstride(m, op->t_in[0]),
                                    (float *)tdev(m, op->t_out), tstride(m, op->t_out), B, op->n);
        case OP_RELU_F32:
            return mhip_relu_f32((const float *)tdev(m, op->t_in[0]), tstride(m, op->t_in[0]), (float *)tdev(m, op->t_out),
                                 tstride(m, op->t_out), B, op->n, op->f0);
        case OP_BINARY_F32:
            return mhip_binary_f32(op->is_mul ? 1 : 0, (const float *)tdev(m, op->t_in[0]), tstride(m, op->t_in[0]),
                                   (const float *)tdev(m, op->t_in[1]), tstride(m, op->t_in[1]),
                                   (float *)tdev(m, op->t_out), tstride(m, op->t_out), B, op->n);
        case OP_BN: {
            const float *s = op->s_off != NO_OFF ? (const float *)(A + op->s_off) : NULL;
            const float *b = op->b_off != NO_OFF ? (const float *)(A + op->b_off) : NULL;
            if (op->is_f32)
                return mhip_batchnorm_f32((const float *)tdev(m, op->t_in[0]), tstride(m, op->t_in[0]),
                                          (float *)tdev(m, op->t_out), tstride(m, op->t_out), B, op->bn_n, op->in_c,
                                          op->in_h * op->in_w, s, b);
            return mhip_batchnorm_i8((const int8_t *)tdev(m, op->t_in[0]), tstride(m, op->t_in[0]),
                                     (int8_t *)tdev(m, op->t_out), tstride(m, op->t_out), B, op->bn_n, op->in_c,
                                     op->in_h * op->in_w, s, b, op->f0, op->f1);
        }
        case OP_MAXPOOL:
            if (op->chain_n) {
                int8_t *outs[3] = {NULL, NULL, NULL};
                size_t strides[3] = {0, 0, 0};
                for (int k = 0; k < op->chain_n; k++) {
                    outs[k] = (int8_t *)tdev(m, op->chain_out[k]);
                    strides[k] = tstride(m, op->chain_out[k]);
                }
                return mhip_pool_chain_i8((const int8_t *)tdev(m, op->t_in[0]), tstride(m, op->t_in[0]), outs, strides,
                                          op->chain_n, B, op->in_h, op->in_w, op->in_c, op->kh, op->kw);
            }
            return mhip_maxpool_i8((const int8_t *)tdev(m, op->t_in[0]), tstride(m, op->t_in[0]), (int8_t *)tdev(m, op->t_out),
                                   tstride(m, op->t_out), B, op->in_h, op->in_w, op->in_c, op->out_h, op->out_w, op->kh,
                                   op->kw, op->sh, op->sw, op->out_pix_stride, op->out_ch_off);
        case OP_CONCAT_SLICE:
            return mhip_concat_slice((const int8_t *)tdev(m, op->t_in[0]), tstride(m, op->t_in[0]),
                                     (int8_t *)tdev(m, op->t_out), tstride(m, op->t_out), B, op->out_h, op->out_w,
                                     op->in_c, op->out_c, op->ch_off);
        case OP_UPSAMPLE_Q: {
            const mars_tensor_t *id = &m->pub.tensors[op->t_in[0]].desc, *od = &m->pub.tensors[op->t_out].desc;
            return mhip_upsample_nchwq((const int8_t *)tdev(m, op->t_in[0]), tstride(m, op->t_in[0]), id->shape[1], id->shape[2], id->shape[3],
                                       (int8_t *)tdev(m, op->t_out), tstride(m, op->t_out), od->shape[1], od->shape[2], od->shape[3], B, op->in_h, op->in_w,
                                       op->in_c, op->out_h, op->out_w, op->scale_h, op->scale_w);
        }
        case OP_MAXPOOL_Q: {
            const mars_tensor_t *id = &m->pub.tensors[op->t_in[0]].desc;
            return mhip_maxpool_nchwq((const int8_t *)tdev(m, op->t_in[0]), tstride(m, op->t_in[0]), (int8_t *)tdev(m, op->t_out), tstride(m, op->t_out), B,
                                      id->shape[1], id->shape[2], id->shape[3], op->kh, op->kw);
        }
        case OP_CONCAT_Q: {
            const int8_t *ins[4];
            size_t strides[4];
            int cs[4];
            for (int k = 0; k < op->n_in && k < 4; k++) {
                ins[k] = (const int8_t *)tdev(m, op->t_in[k]);
                strides[k] = tstride(m, op->t_in[k]);
                cs[k] = m->mt[op->t_in[k]].nhwc_c;
            }
            return mhip_concat_nchwq(ins, strides, cs, op->n_in, (int8_t *)tdev(m, op->t_out), tstride(m, op->t_out), B, op->out_c, op->in_h, op->in_w,
                                     op->rows_only);
        }
        case OP_UPSAMPLE:
            return mhip_upsample_i8((const int8_t *)tdev(m, op->t_in[0]), tstride(m, op->t_in[0]),
                                    (int8_t *)tdev(m, op->t_out), tstride(m, op->t_out), B, op->in_h, op->in_w, op->in_c,
                                    op->out_h, op->out_w, op->scale_h, op->scale_w, op->out_pix_stride, op->out_ch_off);
        default: return -1;
    }
}

static mars_error_t enqueue_plan(mars_model_t *model);
static double now_us(void);
static int tune_raw(const char *key, int value, int *get);

/* a model's tuning overrides in force / taken back (nested calls count: mars_run -> run_device_async) */
static void tune_push(mars_model_ext_t *m) {
    if (m->tune_depth++ || !m->n_tune) return;
    for (int i = 0; i < m->n_tune; i++) {
        tune_raw(m->tune[i].key, 0, &m->tune[i].saved);
        tune_raw(m->tune[i].key, m->tune[i].value, NULL);
    }
}
static void tune_pop(mars_model_ext_t *m) {
    if (--m->tune_depth || !m->n_tune) return;
    for (int i = m->n_tune - 1; i >= 0; i--) tune_raw(m->tune[i].key, m->tune[i].saved, NULL);
}
static mars_error_t run_device_async(mars_model_t *model);
mars_error_t mars_hip_run_device_async(mars_model_t *model) {
    if (!model) return MARS_ERR_INVALID_FILE;
    tune_push((mars_model_ext_t *)model);
    mars_error_t e = run_device_async(model);
    tune_pop((mars_model_ext_t *)model);
    return e;
}

static mars_error_t run_device_async(mars_model_t *model) {
    mars_model_ext_t *m = (mars_model_ext_t *)model;
    if (!m->act_dev || !m->arena_dev) return MARS_ERR_NNA_INIT_FAILED;
    /* graph path: small batch, no per-launch events, buffers not being swapped.  A detection tail still running on the
     * auxiliary stream (mars_hip_detect_device) does not rule it out: the whole graph is ordered behind it (below) */
    int graphable = g_graph_max_batch > 0 && m->batch <= g_graph_max_batch && !m->profiling && !m->pipe;
    for (int i = 0; i < m->n_ops && graphable; i++)
        if (m->ops[i].kind == OP_FAIL) graphable = 0;
    if (!graphable) return enqueue_plan(model);
    if (m->graph_exec && m->graph_gen != g_tune_gen) {
        drop_graph(m);
        m->ran_plain = 0; /* a new launch policy may want workspaces: their first use must not fall inside a capture */
    }
    if (!m->graph_exec) {
        if (!m->ran_plain) { /* first run at this batch: launch by launch (one-time set-up of every launcher happens here) */
            mars_error_t e = enqueue_plan(model);
            if (e == MARS_OK) m->ran_plain = 1;
            return e;
        }
        if (m->tail_pending) { /* the hand-off to a running tail is an outside event: it must not end up inside the capture */
            mhip_select_stream(0);
            if (mhip_stream_wait(0, m->ev_tail_done)) return MARS_ERR_LAYER_FAILED;
            m->tail_pending = 0;
        }
        if (mhip_graph_begin() == 0) {
            mars_error_t e = enqueue_plan(model);
            m->graph_exec = mhip_graph_end(e == MARS_OK);
            m->graph_gen = g_tune_gen;
            if (e != MARS_OK) { m->graph_exec = NULL; return e; }
        }
        VLOG("plan of %d launches at batch %d %s\n", m->n_ops, m->batch, m->graph_exec ? "captured into a HIP graph" : "could not be captured");
        if (!m->graph_exec) { /* capture refused: stay on the plain path for this plan */
            g_graph_max_batch = 0;
            return enqueue_plan(model);
        }
    }
    if (m->tail_pending) { /* the previous run's tail still reads the graph outputs: the replay (all of it) comes after */
        mhip_select_stream(0);
        if (mhip_stream_wait(0, m->ev_tail_done)) return MARS_ERR_LAYER_FAILED;
        m->tail_pending = 0;
    }
    if (mhip_graph_launch(m->graph_exec)) return MARS_ERR_LAYER_FAILED;
    for (uint32_t i = 0; i < model->header.num_layers; i++) model->layers[i].is_executed = true;
    return MARS_OK;
}

/* Enqueue every launch of the plan for frames [m->frame0, m->frame0 + m->run_frames) on stream `sid` (made current). */
static mars_error_t enqueue_range(mars_model_ext_t *m, int sid, int wait_tail) {
    void *prof_last = NULL;
    mhip_select_stream(sid);
    m->scratch_t = -1;
    for (int i = 0; i < m->n_ops; i++) {
        mars_op_t *op = &m->ops[i];
        if (op->kind == OP_FAIL) {
            fprintf(stderr, "Mars: Layer %d execution failed\n", op->layer);
            return (mars_error_t)op->err;
        }
        mars_op_t *mate = op->pair_next && i + 1 < m->n_ops ? &m->ops[i + 1] : NULL;
        if (wait_tail && ((op->t_out >= 0 && m->mt[op->t_out].io_out) || (mate && mate->t_out >= 0 && m->mt[mate->t_out].io_out))) {
            /* the previous batch's detection tail (auxiliary stream) still reads the graph
             * outputs: order this launch behind it */
            mhip_stream_wait(sid, m->ev_tail_done);
            wait_tail = 0;
        }
        if (m->profiling) { /* one event per launch: its stop event is the next launch's start event */
            if (!op->ev1) op->ev1 = mhip_event_create();
            if (!prof_last) {
                if (!op->ev0) op->ev0 = mhip_event_create();
                mhip_event_record(op->ev0);
                prof_last = op->ev0;
            }
            op->ev_start = prof_last;
        }
        int rc;
        if (mate && op->kind == OP_CONV_F32) { /* one grid for both (conv_f32_split's pair form); -2: one after the other */
            mhip_conv_f32_t pa, pb;
            conv_f32_params(m, op, &pa);
            conv_f32_params(m, mate, &pb);
            rc = mhip_conv_f32_pair(&pa, &pb);
            if (rc == -2) {
                rc = launch_op(m, op);
                if (!rc) rc = launch_op(m, mate);
            }
            i++;
        } else if (mate) { /* one grid for both (conv_i8_persist<PAIR>); -2 = not possible at this batch: one after the other */
            mhip_conv_i8_t pa, pb;
            conv_i8_params(m, op, &pa);
            conv_i8_params(m, mate, &pb);
            rc = mhip_conv_i8_pair(&pa, &pb);
            if (rc == -2) {
                rc = launch_op(m, op);
                if (!rc) rc = launch_op(m, mate);
            }
            i++; /* the mate has run */
        } else {
            rc = launch_op(m, op);
        }
        { /* the relayout scratch's copy of a tensor dies with any write to that tensor */
            const mars_op_t *w2[2] = {op, mate};
            for (int q = 0; q < 2; q++)
                if (w2[q]) {
                    if (w2[q]->t_out == m->scratch_t) m->scratch_t = -1;
                    for (int k = 0; k < w2[q]->chain_n; k++)
                        if (w2[q]->chain_out[k] == m->scratch_t) m->scratch_t = -1;
                }
        }
        if (m->profiling) { /* level 2: one event per run of launches of the same kind (their sum lands on the last one) */
            const int nx = i + 1;
            const int end = m->profiling != 2 || nx >= m->n_ops || m->ops[nx].prof_kind != op->prof_kind;
            op->prof_rec = end;
            if (end) {
                mhip_event_record(op->ev1);
                prof_last = op->ev1;
            }
        }
        if (rc != 0) {
            fprintf(stderr, "Mars: Layer %d launch failed: %s\n", op->layer, mhip_last_error());
            return MARS_ERR_LAYER_FAILED;
        }
    }
    return MARS_OK;
}

static mars_error_t enqueue_plan(mars_model_t *model) {
    mars_model_ext_t *m = (mars_model_ext_t *)model;
    for (uint32_t i = 0; i < model->header.num_layers; i++) model->layers[i].is_executed = false;
    const int B = m->batch;
    /* two halves on two streams: not while per-launch events are wanted (profiling), nor inside a graph capture */
    int dual = g_dual_min_batch > 0 && B >= g_dual_min_batch && B >= 2 && !m->profiling &&
               !(g_graph_max_batch > 0 && B <= g_graph_max_batch);
    if (dual) {
        if (!m->ev_fork) m->ev_fork = mhip_event_create_sync();
        for (int k = 0; k < 3; k++) {
            if (!m->ev_join[k]) m->ev_join[k] = mhip_event_create_sync();
            if (!m->ev_join[k]) dual = 0;
        }
        if (!m->ev_fork) dual = 0;
    }
    mars_error_t e;
    if (!dual) {
        m->frame0 = 0; m->run_frames = B;
        e = enqueue_range(m, 0, m->tail_pending);
    } else {
        /* everything the main stream was given before this run (uploads, an earlier run) comes first for every part */
        const int ways = g_dual_ways < B ? g_dual_ways : B;
        mhip_select_stream(0);
        int rc = mhip_event_record(m->ev_fork);
        for (int k = 1; k < ways && !rc; k++) rc = mhip_stream_wait(3 + k, m->ev_fork);
        if (rc) return MARS_ERR_LAYER_FAILED;
        e = MARS_OK;
        int f0 = 0;
        for (int k = 0; k < ways && e == MARS_OK; k++) {
            const int n = (B - f0 + (ways - k) - 1) / (ways - k);
            m->frame0 = f0; m->run_frames = n;
            e = enqueue_range(m, k ? 3 + k : 0, m->tail_pending);
            f0 += n;
        }
        /* join even after a failure: nothing may be left running behind the main stream's back */
        for (int k = 1; k < ways; k++) {
            mhip_select_stream(3 + k);
            rc = mhip_event_record(m->ev_join[k - 1]);
            if (!rc) rc = mhip_stream_wait(0, m->ev_join[k - 1]);
            if (rc && e == MARS_OK) e = MARS_ERR_LAYER_FAILED;
        }
    }
    mhip_select_stream(0);
    m->frame0 = 0; m->run_frames = B;
    if (e != MARS_OK) return e;
    m->tail_pending = 0; /* every output-writing launch above was ordered behind the tail */
    for (uint32_t i = 0; i < model->header.num_layers; i++) model->layers[i].is_executed = true;
    return MARS_OK;
}

static double now_us(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

mars_error_t mars_hip_run_device(mars_model_t *model) {
    double t0 = now_us();
    mars_error_t e = mars_hip_run_device_async(model);
    if (e != MARS_OK) { mhip_sync(); return e; }
    if (mhip_sync()) return MARS_ERR_LAYER_FAILED;
    mars_model_ext_t *m = (mars_model_ext_t *)model;
    if (m->profiling)
        for (int i = 0; i < m->n_ops; i++)
            m->ops[i].last_ms = (m->ops[i].prof_rec && m->ops[i].ev_start && m->ops[i].ev1) ? mhip_event_elapsed_ms(m->ops[i].ev_start, m->ops[i].ev1) : 0.f;
    model->total_inference_us += (uint64_t)(now_us() - t0);
    model->inference_count++;
    return MARS_OK;
}

/* host -> HBM copies of frames [f0, f0 + n) of every graph input, enqueued on the current stream (no synchronisation) */
static mars_error_t enqueue_upload_frames(mars_model_ext_t *m, int f0, int n) {
    for (uint32_t i = 0; i < m->pub.header.num_tensors; i++) {
        mtensor_t *t = &m->mt[i];
        if (!t->io_in || !t->host || !t->dev || t->bytes == 0) continue;
        if (mhip_h2d_2d_async(t->dev + (size_t)f0 * t->stride, t->stride, (const uint8_t *)t->host + (size_t)f0 * t->bytes, t->bytes,
                              t->bytes, (size_t)n))
            return MARS_ERR_LAYER_FAILED;
    }
    return MARS_OK;
}
static mars_error_t enqueue_upload(mars_model_ext_t *m) { return enqueue_upload_frames(m, 0, m->batch); }

/* HBM -> host copies of frames [f0, f0 + n) of every graph output, enqueued on the current stream */
static mars_error_t enqueue_download_frames(mars_model_ext_t *m, int f0, int n) {
    for (uint32_t i = 0; i < m->pub.header.num_tensors; i++) {
        mtensor_t *t = &m->mt[i];
        if (!t->io_out || !t->host || !t->dev || t->bytes == 0) continue;
        uint8_t *host = (uint8_t *)t->host + (size_t)f0 * t->bytes;
        if (t->pix_stride) { /* padded pixel rows (pad_output_rows): frames are exactly pixels x pitch; packed on the
                              * device (a 2-D copy of millions of 255-byte rows runs at a few MB/s), then one copy */
            const size_t rows = (t->bytes / (size_t)t->pix_c) * (size_t)n;
            if (!t->dense_dev || t->stride != (t->bytes / (size_t)t->pix_c) * (size_t)t->pix_stride) return MARS_ERR_LAYER_FAILED;
            uint8_t *dense = (uint8_t *)t->dense_dev + (size_t)f0 * t->bytes;
            if (mhip_unpad_rows(t->dev + (size_t)f0 * t->stride, dense, rows, t->pix_c, t->pix_stride) ||
                mhip_d2h_async(host, dense, t->bytes * (size_t)n))
                return MARS_ERR_LAYER_FAILED;
            continue;
        }
        if (mhip_d2h_2d_async(host, t->bytes, t->dev + (size_t)f0 * t->stride, t->stride, t->bytes, (size_t)n)) return MARS_ERR_LAYER_FAILED;
    }
    return MARS_OK;
}
static mars_error_t enqueue_download(mars_model_ext_t *m) { return enqueue_download_frames(m, 0, m->batch); }

mars_error_t mars_hip_upload_inputs(mars_model_t *model) {
    if (!model) return MARS_ERR_INVALID_FILE;
    mars_error_t e = enqueue_upload((mars_model_ext_t *)model);
    if (mhip_sync() && e == MARS_OK) e = MARS_ERR_LAYER_FAILED;
    return e;
}

mars_error_t mars_hip_download_outputs(mars_model_t *model) {
    if (!model) return MARS_ERR_INVALID_FILE;
    mars_error_t e = enqueue_download((mars_model_ext_t *)model);
    if (mhip_sync() && e == MARS_OK) e = MARS_ERR_LAYER_FAILED;
    return e;
}

/* mars_run at large batches: frames are independent, so the batch goes through in chunks -- chunk k+1 is copied in (upload
 * stream) while chunk k runs (main stream) and chunk k-1 is copied out (download stream).  The caller still gets one
 * synchronous call; the link is busy in both directions nearly all of the time instead of a third of it. */
/* Round 3, traced (rocprofv3 --kernel-trace --memory-copy-trace): smaller chunks lose because a 32- or 64-frame graph is
 * launch-bound (1.2 ms per 32 frames = 9.8 ms of graph for 256 frames against 4.6 ms in one piece), not because of the
 * hand-offs: ordering-only events, all uploads queued up front and copies executed as kernels (mapped host memory) each
 * left the rate where it was or lowered it.  With 2 x 128 frames the return copy (550 MB, 10.3 ms) stays the long pole:
 * 14 k images/s against an ideal 16.3 k for this split; callers that do not need the raw heads switch the copy off
 * (mars_hip_set_output_mode: 24 k images/s, the upload's rate) or use the pipelined calls (mars_pipe.c). */
static int g_run_chunk = 128; /* frames per chunk; batches below twice this go as one piece (tuning key "run_chunk", 0 = never).
                               * Measured, yolov5s twin: batch 256 12.2k -> 14.6k img/s, batch 512 12.3k -> 17.4k; smaller chunks lose
                               * again (each chunk's hand-off between the three streams costs about a millisecond) */
static mars_error_t run_chunked(mars_model_ext_t *m) {
    mars_model_t *model = &m->pub;
    const int B = m->batch;
    int nch = (B + g_run_chunk - 1) / g_run_chunk;
    if (nch > 8) nch = 8;
    for (int k = 0; k < 2; k++)
        for (int c = 0; c < nch; c++)
            if (!m->ev_chunk[k][c] && !(m->ev_chunk[k][c] = mhip_event_create_sync())) return MARS_ERR_ALLOC_FAILED;
    for (uint32_t i = 0; i < model->header.num_layers; i++) model->layers[i].is_executed = false;
    mars_error_t e = MARS_OK;
    /* the copies may not overtake what the caller put on the main stream before this call */
    mhip_select_stream(0);
    if (mhip_event_record(m->ev_chunk[1][nch - 1]) || mhip_stream_wait(2, m->ev_chunk[1][nch - 1])) e = MARS_ERR_LAYER_FAILED;
    /* Every upload is queued FIRST, all of them, then the graphs with their downloads.  The copy engines take their commands
     * in submission order whatever stream they came from: with upload c+1 queued behind download c (the round-2 order:
     * upload, graph, download per chunk) it could not start before download c did, i.e. before graph c had finished --
     * traced: uploads 3.7 ms apart for 1.4 ms of copying each, 13.9 k images/s.  Uploads depend on nothing, so up front they
     * stream back to back and every graph finds its frames waiting. */
    int f0 = 0;
    mhip_select_stream(2);
    for (int c = 0; c < nch && e == MARS_OK; c++) {
        const int n = (B - f0 + (nch - c) - 1) / (nch - c);
        e = enqueue_upload_frames(m, f0, n);
        if (e == MARS_OK && mhip_event_record(m->ev_chunk[0][c])) e = MARS_ERR_LAYER_FAILED;
        f0 += n;
    }
    f0 = 0;
    for (int c = 0; c < nch && e == MARS_OK; c++) {
        const int n = (B - f0 + (nch - c) - 1) / (nch - c);
        mhip_select_stream(0);
        if (mhip_stream_wait(0, m->ev_chunk[0][c])) e = MARS_ERR_LAYER_FAILED;
        if (e == MARS_OK) {
            m->frame0 = f0; m->run_frames = n;
            e = enqueue_range(m, 0, c == 0 ? m->tail_pending : 0);
            mhip_select_stream(0);
        }
        if (e == MARS_OK && mhip_event_record(m->ev_chunk[1][c])) e = MARS_ERR_LAYER_FAILED;
        if (e == MARS_OK && mhip_stream_wait(3, m->ev_chunk[1][c])) e = MARS_ERR_LAYER_FAILED;
        mhip_select_stream(3);
        if (e == MARS_OK && !m->no_download) e = enqueue_download_frames(m, f0, n);
        f0 += n;
    }
    mhip_select_stream(0);
    m->frame0 = 0; m->run_frames = B;
    if (e == MARS_OK) {
        m->tail_pending = 0;
        for (uint32_t i = 0; i < model->header.num_layers; i++) model->layers[i].is_executed = true;
    }
    return e;
}

/* The reference's call: copy in, run, copy out -- synchronous for the caller, but one stream-ordered sequence with ONE
 * synchronisation at its end (three of them cost a single frame 0.05 ms of its 0.7) */
static mars_error_t run_whole(mars_model_t *model);
mars_error_t mars_run(mars_model_t *model) {
    if (!model) return MARS_ERR_INVALID_FILE; /* reference :440 */
    tune_push((mars_model_ext_t *)model);
    mars_error_t e = run_whole(model);
    tune_pop((mars_model_ext_t *)model);
    return e;
}
static mars_error_t run_whole(mars_model_t *model) {
    mars_model_ext_t *m = (mars_model_ext_t *)model;
    if (!m->act_dev || !m->arena_dev) return MARS_ERR_NNA_INIT_FAILED;
    const double t0 = now_us();
    mars_error_t e;
    int chunked = g_run_chunk > 0 && m->batch >= 2 * g_run_chunk && !m->profiling && !m->pipe;
    for (int i = 0; i < m->n_ops && chunked; i++)
        if (m->ops[i].kind == OP_FAIL) chunked = 0; /* a failing layer: the plain path reports it the reference's way */
    if (chunked) {
        e = run_chunked(m);
    } else {
        e = enqueue_upload(m);
        if (e == MARS_OK) e = mars_hip_run_device_async(model);
        if (e == MARS_OK && !m->no_download) e = enqueue_download(m);
    }
    if (mhip_sync() && e == MARS_OK) e = MARS_ERR_LAYER_FAILED;
    if (e != MARS_OK) return e;
    if (m->profiling)
        for (int i = 0; i < m->n_ops; i++)
            m->ops[i].last_ms = (m->ops[i].prof_rec && m->ops[i].ev_start && m->ops[i].ev1) ? mhip_event_elapsed_ms(m->ops[i].ev_start, m->ops[i].ev1) : 0.f;
    model->total_inference_us += (uint64_t)(now_us() - t0);
    model->inference_count++;
    return MARS_OK;
}

mars_error_t mars_hip_sync(void) { return mhip_sync() ? MARS_ERR_LAYER_FAILED : MARS_OK; }


/* --------------------------------------------------------------- extensions */
mars_error_t mars_hip_set_batch(mars_model_t *model, int n) {
    if (!model || n <= 0 || n > 65535) return MARS_ERR_INVALID_FILE;
    return alloc_batch((mars_model_ext_t *)model, n);
}

int mars_hip_get_batch(const mars_model_t *model) { return model ? ((const mars_model_ext_t *)model)->batch : 0; }

mars_error_t mars_hip_set_output_mode(mars_model_t *model, int mode) {
    if (!model || (mode != MARS_HIP_OUTPUT_HEADS && mode != MARS_HIP_OUTPUT_ON_DEVICE)) return MARS_ERR_INVALID_FILE;
    ((mars_model_ext_t *)model)->no_download = mode == MARS_HIP_OUTPUT_ON_DEVICE;
    return MARS_OK;
}

mars_error_t mars_hip_set_fusion(mars_model_t *model, int level) {
    if (!model) return MARS_ERR_INVALID_FILE;
    mars_model_ext_t *m = (mars_model_ext_t *)model;
    if (m->deferred) return MARS_ERR_INVALID_FILE;
    mhip_sync();
    m->fusion = level;
    mars_error_t e = build_plan(m);
    if (e == MARS_OK) e = upload_params(m);
    if (e == MARS_OK) e = alloc_batch(m, m->batch > 0 ? m->batch : 1);
    return e;
}

/* Launch-policy knobs.  Host-side ones live here, the convolution's in conv_i8.hip; tune_raw sets or reads one without
 * touching the graph generation (the per-model overrides below go through it around every run). */
static int tune_raw(const char *key, int value, int *get) {
    if (!key) return -1;
    struct { const char *k; int *v; int lo, hi; } tab[] = {
        {"graph_max_batch", &g_graph_max_batch, 0, 1 << 30},      /* largest batch whose plan is replayed as a HIP graph (0 = never) */
        {"dual_stream_min_batch", &g_dual_min_batch, 0, 1 << 30}, /* smallest batch that runs as two halves on two streams (0 = never) */
        {"run_chunk", &g_run_chunk, 0, 1 << 30}, /* mars_run: frames per overlapped chunk at batches of at least twice this (0 = never) */
        {"dual_stream_ways", &g_dual_ways, 2, 4},
    };
    for (size_t i = 0; i < sizeof tab / sizeof tab[0]; i++)
        if (!strcmp(key, tab[i].k)) {
            if (get) { *get = *tab[i].v; return 0; }
            if (value < tab[i].lo || value > tab[i].hi) return -1;
            *tab[i].v = value;
            return 0;
        }
    if (!strcmp(key, "f32_mfma")) { /* 0 exact everywhere, 1 f32 matrix cores where provably safe (default), 2 everywhere, 3 / 4 bf16 matrix cores on split operands, three / six piece products */
        if (get) { *get = mhip_conv_f32_mode(-1); return 0; }
        if (value < 0 || value > 4) return -1;
        mhip_conv_f32_mode(value);
        return 0;
    }
    return get ? mhip_conv_i8_tune_get(key, get) : mhip_conv_i8_tune(key, value);
}

/* the plan of a float model depends on the f32_mfma mode it was built under -- which bf16 weight images sit in the arena (two planes
 * under 3, three under 4, conv_f32_patch / conv_f32_stem images under 3 only) and which tensors are kept in record format (rec_pairs, mode 3).
 * When the mode in force for a model changes across the 2 | 3 | 4 boundaries it is planned again, exactly as mars_hip_set_fusion does: parameters
 * re-packed and uploaded, tensors and I/O staging reallocated and zeroed (callers set the mode before they fill inputs).  Descriptor-only ranks
 * (no weight blob to re-pack) keep their plan; conv_f32_params then routes every launch to a kernel whose operands exist. */
static mars_error_t replan_for_f32_mode(mars_model_ext_t *m) {
    if (m->deferred || !m->arena_dev) return MARS_OK;
    int has_f32 = 0;
    for (int i = 0; i < m->n_ops && !has_f32; i++) has_f32 = m->ops[i].kind == OP_CONV_F32;
    if (!has_f32) return MARS_OK;
    tune_push(m);
    const int mode = mhip_conv_f32_mode(-1);
    mars_error_t e = MARS_OK;
    if (mode != m->plan_f32_mode && (mode >= 3 || m->plan_f32_mode >= 3)) {
        mhip_sync();
        e = build_plan(m);
        if (e == MARS_OK) e = upload_params(m);
        if (e == MARS_OK) e = alloc_batch(m, m->batch > 0 ? m->batch : 1);
    }
    tune_pop(m);
    return e;
}

int mars_hip_set_tuning(const char *key, int value) {
    g_tune_gen++; /* captured graphs froze the launch policy they were recorded under */
    int rc = tune_raw(key, value, NULL);
    if (rc == 0 && key && !strcmp(key, "f32_mfma"))
        for (mars_model_ext_t *m = mars_live_models(); m; m = m->live_next)
            if (replan_for_f32_mode(m) != MARS_OK) rc = -1;
    return rc;
}

int mars_hip_get_tuning(const char *key, int *value) { return value ? tune_raw(key, 0, value) : -1; }

/* Per-model overrides: kept on the model, put in force for the duration of each of ITS runs (tune_push / tune_pop around
 * mars_run, mars_hip_run_device_async, mars_hip_autotune) and taken back afterwards, so two models in one process can run
 * under different policies while mars_hip_set_tuning stays the process default. */
int mars_hip_model_set_tuning(mars_model_t *model, const char *key, int value) {
    mars_model_ext_t *m = (mars_model_ext_t *)model;
    int cur;
    if (!m || !key || strlen(key) >= sizeof m->tune[0].key || tune_raw(key, 0, &cur)) return -1;
    if (tune_raw(key, value, NULL)) return -1; /* validates the value ... */
    tune_raw(key, cur, NULL);                  /* ... without leaving it in force */
    int i = 0;
    while (i < m->n_tune && strcmp(m->tune[i].key, key)) i++;
    if (i == m->n_tune) {
        if (m->n_tune == MARS_MAX_MODEL_TUNE) return -1;
        strcpy(m->tune[m->n_tune++].key, key);
    }
    m->tune[i].value = value;
    drop_graph(m); /* its captured graph froze the old policy */
    if (!strcmp(key, "f32_mfma") && replan_for_f32_mode(m) != MARS_OK) return -1;
    return 0;
}

int mars_hip_model_get_tuning(mars_model_t *model, const char *key, int *value) {
    mars_model_ext_t *m = (mars_model_ext_t *)model;
    if (!m || !key || !value) return -1;
    for (int i = 0; i < m->n_tune; i++)
        if (!strcmp(m->tune[i].key, key)) { *value = m->tune[i].value; return 0; }
    return tune_raw(key, 0, value); /* not overridden: the process default */
}

/* Time every launch variant of every int8 convolution on the device, at the current batch, and pin the fastest
 * (all variants write the same bytes; the layer's real buffers are used, so the tensors stay valid). */
static mars_error_t autotune_model(mars_model_t *model, int reps);
mars_error_t mars_hip_autotune(mars_model_t *model, int reps) {
    if (!model) return MARS_ERR_INVALID_FILE;
    tune_push((mars_model_ext_t *)model);
    mars_error_t e = autotune_model(model, reps);
    tune_pop((mars_model_ext_t *)model);
    return e;
}
static mars_error_t autotune_model(mars_model_t *model, int reps) {
    mars_model_ext_t *m = (mars_model_ext_t *)model;
    if (!m->act_dev || !m->arena_dev) return MARS_ERR_NNA_INIT_FAILED;
    drop_graph(m);
    if (reps <= 0) reps = 3;
    void *e0 = mhip_event_create(), *e1 = mhip_event_create();
    if (!e0 || !e1) return MARS_ERR_ALLOC_FAILED;
    mars_error_t err = MARS_OK;
    if (mhip_sync()) err = MARS_ERR_LAYER_FAILED;
    for (int i = 0; i < m->n_ops && err == MARS_OK; i++) {
        mars_op_t *op = &m->ops[i];
        if (op->kind != OP_CONV_I8 || op->nchw) continue;
        if (op->pair_next || (i > 0 && m->ops[i - 1].pair_next)) continue; /* paired launches have one form */
        mhip_conv_i8_t p;
        conv_i8_params(m, op, &p);
        int codes[32];
        const int n = mhip_conv_i8_variants(&p, codes, 32);
        if (getenv("MARS_VERBOSE") && atoi(getenv("MARS_VERBOSE")) > 1)
            fprintf(stderr, "Mars: autotune layer %d: %dx%dx%d -> %dx%dx%d k%dx%d s%d pixstride %d choff %d lut %d safe %d\n", op->layer,
                    p.in_h, p.in_w, p.in_c, p.out_h, p.out_w, p.out_c, p.kh, p.kw, p.stride_w, p.out_pix_stride, p.out_ch_off,
                    p.lut != NULL, p.safe);
        float best = 0.0f;
        int best_code = 0;
        for (int k = 0; k < n && err == MARS_OK; k++) {
            p.variant = codes[k];
            int rc = mhip_conv_i8(&p); /* warm: code object load, occupancy query */
            if (!rc) rc = mhip_event_record(e0);
            for (int r = 0; r < reps && !rc; r++) rc = mhip_conv_i8(&p);
            if (!rc) rc = mhip_event_record(e1);
            if (rc || mhip_sync()) { err = MARS_ERR_LAYER_FAILED; break; }
            const float ms = mhip_event_elapsed_ms(e0, e1);
            if (getenv("MARS_VERBOSE") && atoi(getenv("MARS_VERBOSE")) > 1)
                fprintf(stderr, "Mars: autotune layer %d: candidate %d: %.1f us\n", op->layer, codes[k], ms * 1000.0f / reps);
            if (best_code == 0 || ms < best) { best = ms; best_code = codes[k]; }
        }
        if (err == MARS_OK && best_code) {
            if (getenv("MARS_VERBOSE"))
                fprintf(stderr, "Mars: autotune layer %d: variant %d (%.1f us) of %d candidates, default %d\n", op->layer,
                        best_code, best * 1000.0f / reps, n, n ? codes[0] : 0);
            op->variant = best_code;
        }
    }
    mhip_event_destroy(e0);
    mhip_event_destroy(e1);
    return err;
}

void *mars_hip_tensor_device(mars_model_t *model, int ti, size_t *frame_stride) {
    if (!model || ti < 0 || (uint32_t)ti >= model->header.num_tensors) return NULL;
    mars_model_ext_t *m = (mars_model_ext_t *)model;
    if (frame_stride) *frame_stride = m->mt[ti].stride;
    return m->mt[ti].dev;
}

size_t mars_hip_tensor_frame_bytes(const mars_model_t *model, int ti) {
    if (!model || ti < 0 || (uint32_t)ti >= model->header.num_tensors) return 0;
    return ((const mars_model_ext_t *)model)->mt[ti].bytes;
}

int mars_hip_tensor_row_pitch(mars_model_t *model, int ti, int *row_bytes) {
    if (!model || ti < 0 || (uint32_t)ti >= model->header.num_tensors) return 0;
    const mtensor_t *t = &((mars_model_ext_t *)model)->mt[ti];
    if (row_bytes) *row_bytes = t->pix_stride ? t->pix_c : 0;
    return t->pix_stride;
}

mars_error_t mars_hip_read_tensor(mars_model_t *model, int ti, int frame, void *dst, size_t bytes) {
    if (!model || !dst || ti < 0 || (uint32_t)ti >= model->header.num_tensors) return MARS_ERR_INVALID_TENSOR;
    mars_model_ext_t *m = (mars_model_ext_t *)model;
    mtensor_t *t = &m->mt[ti];
    if (!t->dev || frame < 0 || (!t->is_weight && frame >= m->batch)) return MARS_ERR_INVALID_TENSOR;
    if (!t->is_weight && bytes > t->stride) return MARS_ERR_INVALID_TENSOR;
    if (t->partial) return MARS_ERR_INVALID_TENSOR; /* (like a tensor a fusion pass elided: only the rows its reader needs exist) */
    if (t->is_weight) { /* weights: one copy for every frame, bounded by the blob mirror */
        const size_t off = (size_t)m->pub.tensors[ti].desc.data_offset;
        if (off > m->blob_mirror_bytes || bytes > m->blob_mirror_bytes - off) return MARS_ERR_INVALID_TENSOR;
        frame = 0;
    }
    if (mhip_sync()) return MARS_ERR_LAYER_FAILED;
    if (t->rec_c) { /* record format (rec_pairs): the frame comes back as the NCHW floats hi + mid (the value to 2^-16) */
        const size_t full = (size_t)t->rec_c * t->rec_hw * 4;
        uint8_t *raw = (uint8_t *)malloc(full);
        float *f = (float *)malloc(full);
        if (!raw || !f || bytes > full || mhip_d2h_async(raw, t->dev + (size_t)frame * t->stride, full) || mhip_sync()) {
            free(raw); free(f);
            return MARS_ERR_LAYER_FAILED;
        }
        for (int c = 0; c < t->rec_c; c++)
            for (int px = 0; px < t->rec_hw; px++) {
                const uint8_t *r = raw + ((size_t)(c >> 3) * t->rec_hw + px) * 32 + (size_t)(c & 7) * 2;
                uint16_t h, md;
                memcpy(&h, r, 2); memcpy(&md, r + 16, 2);
                const uint32_t hb = (uint32_t)h << 16, mb = (uint32_t)md << 16;
                float hf, mf;
                memcpy(&hf, &hb, 4); memcpy(&mf, &mb, 4);
                f[(size_t)c * t->rec_hw + px] = hf + mf;
            }
        memcpy(dst, f, bytes);
        free(raw); free(f);
        return MARS_OK;
    }
    if (t->nhwc_c) { /* kept as pixels x channels (nhwc_internal): the frame comes back in the reference's [C][H][W] order */
        const size_t full = (size_t)t->nhwc_c * t->nhwc_hw, pitch = t->nhwc_pitch ? (size_t)t->nhwc_pitch : (size_t)t->nhwc_c, dev_b = pitch * t->nhwc_hw;
        uint8_t *raw = (uint8_t *)malloc(dev_b), *pl = (uint8_t *)malloc(full);
        if (!raw || !pl || bytes > full || dev_b > t->stride || mhip_d2h_async(raw, t->dev + (size_t)frame * t->stride, dev_b) || mhip_sync()) {
            free(raw); free(pl);
            return MARS_ERR_LAYER_FAILED;
        }
        for (int px = 0; px < t->nhwc_hw; px++)
            for (int c = 0; c < t->nhwc_c; c++) pl[(size_t)c * t->nhwc_hw + px] = raw[(size_t)px * pitch + c];
        memcpy(dst, pl, bytes);
        free(raw); free(pl);
        return MARS_OK;
    }
    if (t->pix_stride) { /* padded pixel rows: the frame is packed on the device first */
        uint8_t *dense = t->dense_dev + (size_t)frame * t->bytes;
        if (bytes > t->bytes || !t->dense_dev) return MARS_ERR_INVALID_TENSOR;
        if (mhip_unpad_rows(t->dev + (size_t)frame * t->stride, dense, t->bytes / (size_t)t->pix_c, t->pix_c, t->pix_stride) ||
            mhip_d2h_async(dst, dense, bytes) || mhip_sync())
            return MARS_ERR_LAYER_FAILED;
        return MARS_OK;
    }
    if (mhip_d2h_async(dst, t->dev + (size_t)frame * t->stride, bytes) || mhip_sync()) return MARS_ERR_LAYER_FAILED;
    return MARS_OK;
}

mars_error_t mars_hip_write_tensor(mars_model_t *model, int ti, int frame, const void *src, size_t bytes) {
    if (!model || !src || ti < 0 || (uint32_t)ti >= model->header.num_tensors) return MARS_ERR_INVALID_TENSOR;
    mars_model_ext_t *m = (mars_model_ext_t *)model;
    mtensor_t *t = &m->mt[ti];
    if (!t->dev || t->is_weight || frame < 0 || frame >= m->batch || bytes > t->stride) return MARS_ERR_INVALID_TENSOR;
    if (t->rec_c) { /* record format: whole frames only, cut into the two bf16 pieces as the producing kernel would */
        const size_t full = (size_t)t->rec_c * t->rec_hw * 4;
        if (bytes != full) return MARS_ERR_INVALID_TENSOR;
        uint8_t *raw = (uint8_t *)calloc(1, full);
        if (!raw) return MARS_ERR_ALLOC_FAILED;
        const float *f = (const float *)src;
        for (int c = 0; c < t->rec_c; c++)
            for (int px = 0; px < t->rec_hw; px++) {
                float x, hv, r;
                memcpy(&x, &f[(size_t)c * t->rec_hw + px], 4);
                uint32_t b;
                memcpy(&b, &x, 4);
                const uint16_t h = (b & 0x7fffffffu) > 0x7f800000u ? (uint16_t)((b >> 16) | 0x40u) : (uint16_t)((b + 0x7fffu + ((b >> 16) & 1u)) >> 16);
                const uint32_t hb = (uint32_t)h << 16;
                memcpy(&hv, &hb, 4);
                r = hv - hv == 0.0f ? x - hv : 0.0f;
                memcpy(&b, &r, 4);
                const uint16_t md = (b & 0x7fffffffu) > 0x7f800000u ? (uint16_t)((b >> 16) | 0x40u) : (uint16_t)((b + 0x7fffu + ((b >> 16) & 1u)) >> 16);
                uint8_t *q = raw + ((size_t)(c >> 3) * t->rec_hw + px) * 32 + (size_t)(c & 7) * 2;
                memcpy(q, &h, 2); memcpy(q + 16, &md, 2);
            }
        const int rc = mhip_h2d_async(t->dev + (size_t)frame * t->stride, raw, full) || mhip_sync();
        free(raw);
        return rc ? MARS_ERR_LAYER_FAILED : MARS_OK;
    }
    if (t->zero_from && bytes > t->zero_from) { /* a convolution skips these bytes because no layer ever writes them: they must stay zero */
        const uint8_t *b = (const uint8_t *)src;
        for (size_t i = t->zero_from; i < bytes; i++)
            if (b[i]) return MARS_ERR_INVALID_TENSOR;
    }
    if (t->nhwc_c) { /* kept as pixels x channels: whole frames only, given in the reference's [C][H][W] order */
        const size_t full = (size_t)t->nhwc_c * t->nhwc_hw, pitch = t->nhwc_pitch ? (size_t)t->nhwc_pitch : (size_t)t->nhwc_c, dev_b = pitch * t->nhwc_hw;
        if (bytes != full || dev_b > t->stride) return MARS_ERR_INVALID_TENSOR;
        uint8_t *raw = (uint8_t *)calloc(1, dev_b);
        if (!raw) return MARS_ERR_ALLOC_FAILED;
        const uint8_t *pl = (const uint8_t *)src;
        for (int px = 0; px < t->nhwc_hw; px++)
            for (int c = 0; c < t->nhwc_c; c++) raw[(size_t)px * pitch + c] = pl[(size_t)c * t->nhwc_hw + px];
        const int rc = mhip_h2d_async(t->dev + (size_t)frame * t->stride, raw, dev_b) || mhip_sync();
        free(raw);
        return rc ? MARS_ERR_LAYER_FAILED : MARS_OK;
    }
    if (t->pix_stride) { /* padded pixel rows: whole pixels only */
        const size_t rows = bytes / (size_t)t->pix_c;
        if (bytes > t->bytes || rows * (size_t)t->pix_c != bytes) return MARS_ERR_INVALID_TENSOR;
        if (mhip_h2d_2d_async(t->dev + (size_t)frame * t->stride, (size_t)t->pix_stride, src, (size_t)t->pix_c, (size_t)t->pix_c, rows) || mhip_sync())
            return MARS_ERR_LAYER_FAILED;
        return MARS_OK;
    }
    if (mhip_h2d_async(t->dev + (size_t)frame * t->stride, src, bytes) || mhip_sync()) return MARS_ERR_LAYER_FAILED;
    return MARS_OK;
}

void mars_hip_set_profiling(mars_model_t *model, int on) {
    if (model) ((mars_model_ext_t *)model)->profiling = on;
}

int mars_hip_num_ops(const mars_model_t *model) { return model ? ((const mars_model_ext_t *)model)->n_ops : 0; }

int mars_hip_op_info(const mars_model_t *model, int i, int *layer, int *kind, double *macs, double *bytes, float *last_ms) {
    if (!model) return -1;
    const mars_model_ext_t *m = (const mars_model_ext_t *)model;
    if (i < 0 || i >= m->n_ops) return -1;
    if (layer) *layer = m->ops[i].layer;
    if (kind) *kind = m->ops[i].prof_kind;
    if (macs) *macs = m->ops[i].macs;
    if (bytes) *bytes = m->ops[i].bytes;
    if (last_ms) { /* events of the most recent (completed) run; waits for the stop event */
        mars_op_t *op = (mars_op_t *)&m->ops[i];
        if (m->profiling) op->last_ms = (op->prof_rec && op->ev_start && op->ev1) ? mhip_event_elapsed_ms(op->ev_start, op->ev1) : 0.f;
        *last_ms = op->last_ms;
    }
    return 0;
}

void *mars_hip_stream(void) { return mhip_stream(); }

void *mars_hip_param_arena(mars_model_t *model, size_t *bytes) {
    if (!model) return NULL;
    mars_model_ext_t *m = (mars_model_ext_t *)model;
    if (bytes) *bytes = m->arena_size;
    return m->arena_dev;
}

