/*
 * mars_pipe.c -- double-buffered host I/O around the graph (additive extension, include/mars_hip.h).
 *
 * The reference's call pattern is synchronous: fill mars_get_input()->vaddr, mars_run(), read
 * mars_get_output()->vaddr (reference src/mars/mars_test.c:73-120).  Through mars_run() a batch pays host->HBM copy,
 * graph and HBM->host copy one after the other on one stream (measured: 12 k images/s against 51 k with resident
 * inputs).  A camera pipeline does not need that order: while batch k is computed, batch k+1 can travel to the device
 * and the results of batch k-1 back.  mars_hip_pipe_* gives the caller exactly that with three batches in flight over
 * four buffer sets (the fourth keeps a collected result readable while the next batch is being queued):
 *
 *     upload stream    H2D(k+1)                 |  graph inputs / outputs of a slot are separate HBM buffers, so the
 *     main stream      graph(k)                 |  three stages never touch the same bytes; events order the hand-offs
 *     aux stream       decode + NMS(k)          |  (upload -> graph -> tail -> download -> slot free)
 *     download stream  D2H(k-1)
 *
 * Camera mode (opts.camera_w / camera_h): what is uploaded for input 0 is the camera's uint8 RGB frames, and the image front-end
 * (letterbox resize + px - 128, mars_preproc.c) runs behind the copy on the upload stream: the reference demo's whole loop.
 *
 * What travels back is the caller's choice: the raw graph outputs (as mars_run leaves them), the detections of the
 * decode + NMS tail (24 KB per frame at most instead of 2.1 MB of head tensors), or both.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mars_internal.h"

#define PIPE_DEPTH 3 /* batches in flight.  With two, batch k+1 could only be queued once batch k-1 had fully come back: its
                        upload then started a millisecond into graph k and graph k+1 waited for it (34 k instead of 45 k
                        images/s measured) */
#define PIPE_SLOTS 4 /* buffer sets: one more than the depth, so that the slot whose results mars_hip_pipe_wait() has just
                        handed out is NOT the one the next submit refills (with three sets it was: the steady-state loop
                        "wait k, submit k+3" queued copies into the buffers the caller was still reading) */

typedef struct {
    uint8_t *in_host[MARS_MAX_IO], *in_dev[MARS_MAX_IO];
    uint8_t *out_host[MARS_MAX_IO], *out_dev[MARS_MAX_IO], *out_dense[MARS_MAX_IO];
    uint8_t *rgb_dev; /* camera mode: the uploaded RGB frames of input 0 (in_host[0] is their pinned source) */
    mars_det_t *det_host;
    int *cnt_host;
    void *det_dev;
    int *cnt_dev;
    void *ev_up, *ev_graph, *ev_tail, *ev_down;
    void *ev_pre0, *ev_pre1; /* camera mode: around the front-end kernel (timing events: mars_hip_pipe_camera_ms) */
    int used; /* the slot has been submitted before: its events are meaningful */
} pipe_slot_t;

typedef struct mars_pipe {
    mars_hip_pipe_opts_t opts;
    pipe_slot_t slot[PIPE_SLOTS];
    int head, tail, inflight;
    uint8_t *saved_in[MARS_MAX_IO], *saved_out[MARS_MAX_IO]; /* the model's own I/O tensor buffers */
    int batch;
} mars_pipe_t;

static mtensor_t *io_tensor(mars_model_ext_t *m, int is_out, int i) {
    const uint32_t n = is_out ? m->pub.header.num_outputs : m->pub.header.num_inputs;
    if (i < 0 || (uint32_t)i >= n) return NULL;
    const uint32_t tid = is_out ? m->pub.header.output_tensor_ids[i] : m->pub.header.input_tensor_ids[i];
    if (tid >= m->pub.header.num_tensors || m->mt[tid].is_weight || !m->mt[tid].dev) return NULL;
    return &m->mt[tid];
}

static void slot_free(pipe_slot_t *s) {
    for (int i = 0; i < MARS_MAX_IO; i++) {
        if (s->in_host[i]) mhip_host_free(s->in_host[i]);
        if (s->in_dev[i]) mhip_free(s->in_dev[i]);
        if (s->out_host[i]) mhip_host_free(s->out_host[i]);
        if (s->out_dev[i]) mhip_free(s->out_dev[i]);
        if (s->out_dense[i]) mhip_free(s->out_dense[i]);
    }
    if (s->rgb_dev) mhip_free(s->rgb_dev);
    if (s->det_host) mhip_host_free(s->det_host);
    if (s->cnt_host) mhip_host_free(s->cnt_host);
    if (s->det_dev) mhip_free(s->det_dev);
    if (s->cnt_dev) mhip_free(s->cnt_dev);
    if (s->ev_up) mhip_event_destroy(s->ev_up);
    if (s->ev_graph) mhip_event_destroy(s->ev_graph);
    if (s->ev_tail) mhip_event_destroy(s->ev_tail);
    if (s->ev_down) mhip_event_destroy(s->ev_down);
    if (s->ev_pre0) mhip_event_destroy(s->ev_pre0);
    if (s->ev_pre1) mhip_event_destroy(s->ev_pre1);
    memset(s, 0, sizeof(*s));
}

void mars_hip_pipe_close(mars_model_t *model) {
    if (!model) return;
    mars_model_ext_t *m = (mars_model_ext_t *)model;
    mars_pipe_t *pp = (mars_pipe_t *)m->pipe;
    if (!pp) return;
    if (mhip_ready()) mhip_sync();
    /* the model's tensors point at their own buffers again (only if the batch was not re-planned meanwhile: then
     * alloc_batch has already given them new ones) */
    if (m->batch == pp->batch && m->act_dev) {
        for (uint32_t i = 0; i < m->pub.header.num_inputs && i < MARS_MAX_IO; i++) {
            mtensor_t *t = io_tensor(m, 0, (int)i);
            if (t && pp->saved_in[i]) t->dev = pp->saved_in[i];
        }
        for (uint32_t i = 0; i < m->pub.header.num_outputs && i < MARS_MAX_IO; i++) {
            mtensor_t *t = io_tensor(m, 1, (int)i);
            if (t && pp->saved_out[i]) t->dev = pp->saved_out[i];
        }
    }
    for (int s = 0; s < PIPE_SLOTS; s++) slot_free(&pp->slot[s]);
    free(pp);
    m->pipe = NULL;
}

mars_error_t mars_hip_pipe_open(mars_model_t *model, const mars_hip_pipe_opts_t *opts) {
    if (!model || !opts) return MARS_ERR_INVALID_FILE;
    mars_model_ext_t *m = (mars_model_ext_t *)model;
    if (!m->act_dev || !m->arena_dev) return MARS_ERR_NNA_INIT_FAILED;
    if (!opts->download_outputs && !opts->detect) return MARS_ERR_INVALID_LAYER; /* nothing would come back */
    if (opts->detect && (opts->n_det_outputs <= 0 || opts->n_det_outputs > 4)) return MARS_ERR_INVALID_TENSOR;
    const int camera = opts->camera_w > 0 && opts->camera_h > 0;
    if ((opts->camera_w > 0) != (opts->camera_h > 0) || opts->camera_w < 0 || opts->camera_h < 0) return MARS_ERR_INVALID_FILE;
    if (model->header.num_inputs > MARS_MAX_IO || model->header.num_outputs > MARS_MAX_IO) return MARS_ERR_INVALID_FILE;
    mars_hip_pipe_close(model);
    if (mhip_sync()) return MARS_ERR_LAYER_FAILED;
    mars_pipe_t *pp = (mars_pipe_t *)calloc(1, sizeof(*pp));
    if (!pp) return MARS_ERR_ALLOC_FAILED;
    pp->opts = *opts;
    pp->batch = m->batch;
    m->pipe = pp;
    const size_t B = (size_t)m->batch;
    mars_error_t err = MARS_OK;
    for (uint32_t i = 0; i < model->header.num_inputs; i++) {
        mtensor_t *t = io_tensor(m, 0, (int)i);
        pp->saved_in[i] = t ? t->dev : NULL;
    }
    for (uint32_t i = 0; i < model->header.num_outputs; i++) {
        mtensor_t *t = io_tensor(m, 1, (int)i);
        pp->saved_out[i] = t ? t->dev : NULL;
    }
    for (int s = 0; s < PIPE_SLOTS && err == MARS_OK; s++) {
        pipe_slot_t *sl = &pp->slot[s];
        for (uint32_t i = 0; i < model->header.num_inputs && err == MARS_OK; i++) {
            mtensor_t *t = io_tensor(m, 0, (int)i);
            if (!t) continue;
            const size_t cam_b = (i == 0 && camera) ? (size_t)opts->camera_w * opts->camera_h * 3 * B : 0;
            const size_t host_b = cam_b ? cam_b : (t->bytes > 0 ? t->bytes * B : 64);
            sl->in_host[i] = (uint8_t *)mhip_host_alloc(host_b);
            sl->in_dev[i] = (uint8_t *)mhip_malloc(t->stride * B + 256);
            if (cam_b) sl->rgb_dev = (uint8_t *)mhip_malloc(cam_b);
            if (!sl->in_host[i] || !sl->in_dev[i] || (cam_b && !sl->rgb_dev)) { err = MARS_ERR_ALLOC_FAILED; break; }
            memset(sl->in_host[i], 0, host_b);
            if (mhip_memset_async(sl->in_dev[i], 0, t->stride * B + 256)) err = MARS_ERR_ALLOC_FAILED;
        }
        for (uint32_t i = 0; i < model->header.num_outputs && err == MARS_OK; i++) {
            mtensor_t *t = io_tensor(m, 1, (int)i);
            if (!t) continue;
            sl->out_dev[i] = (uint8_t *)mhip_malloc(t->stride * B + 256);
            if (!sl->out_dev[i]) { err = MARS_ERR_ALLOC_FAILED; break; }
            if (mhip_memset_async(sl->out_dev[i], 0, t->stride * B + 256)) err = MARS_ERR_ALLOC_FAILED;
            if (opts->download_outputs) {
                sl->out_host[i] = (uint8_t *)mhip_host_alloc(t->bytes > 0 ? t->bytes * B : 64);
                if (!sl->out_host[i]) { err = MARS_ERR_ALLOC_FAILED; break; }
                if (t->pix_stride) {
                    sl->out_dense[i] = (uint8_t *)mhip_malloc(t->bytes * B);
                    if (!sl->out_dense[i]) { err = MARS_ERR_ALLOC_FAILED; break; }
                }
            }
        }
        if (opts->detect && err == MARS_OK) {
            sl->det_host = (mars_det_t *)mhip_host_alloc(B * MARS_YOLO_MAX_DET * sizeof(mars_det_t));
            sl->cnt_host = (int *)mhip_host_alloc(B * sizeof(int));
            sl->det_dev = mhip_malloc(B * MARS_YOLO_MAX_DET * sizeof(mars_det_t));
            sl->cnt_dev = (int *)mhip_malloc(B * 2 * sizeof(int));
            if (!sl->det_host || !sl->cnt_host || !sl->det_dev || !sl->cnt_dev) err = MARS_ERR_ALLOC_FAILED;
        }
        sl->ev_up = mhip_event_create_sync();
        sl->ev_graph = mhip_event_create_sync();
        sl->ev_tail = mhip_event_create_sync();
        sl->ev_down = mhip_event_create();
        if (!sl->ev_up || !sl->ev_graph || !sl->ev_tail || !sl->ev_down) err = MARS_ERR_ALLOC_FAILED;
        if (camera) {
            sl->ev_pre0 = mhip_event_create();
            sl->ev_pre1 = mhip_event_create();
            if (!sl->ev_pre0 || !sl->ev_pre1) err = MARS_ERR_ALLOC_FAILED;
        }
    }
    if (err == MARS_OK && camera) { /* the front-end's gather tables: built and uploaded now, not behind the first submit */
        mtensor_t *t0 = io_tensor(m, 0, 0); /* validates the id: the loader tolerates input ids beyond the tensor table */
        if (!t0) err = MARS_ERR_INVALID_TENSOR;
        else {
            const mars_tensor_t *d0 = &model->tensors[model->header.input_tensor_ids[0]].desc;
            const int nhwc = d0->format == MARS_FORMAT_NHWC;
            const int th = nhwc ? d0->shape[1] : d0->shape[2], tw = nhwc ? d0->shape[2] : d0->shape[3], ch = nhwc ? d0->shape[3] : d0->shape[1];
            if (ch != 3 || d0->dtype != MARS_DTYPE_INT8 || mars_preproc_prepare(opts->camera_w, opts->camera_h, tw, th)) err = MARS_ERR_INVALID_TENSOR;
        }
    }
    if (err == MARS_OK && opts->detect) /* decode tables: built and uploaded once, before anything is in flight */
        err = mars_detect_prepare(m, opts->det_outputs, opts->n_det_outputs);
    if (err == MARS_OK && mhip_sync()) err = MARS_ERR_LAYER_FAILED;
    if (err != MARS_OK) mars_hip_pipe_close(model);
    return err;
}

void *mars_hip_pipe_input(mars_model_t *model, int input_index) {
    if (!model) return NULL;
    mars_pipe_t *pp = (mars_pipe_t *)((mars_model_ext_t *)model)->pipe;
    if (!pp || input_index < 0 || input_index >= MARS_MAX_IO) return NULL;
    return pp->slot[pp->head].in_host[input_index];
}

mars_error_t mars_hip_pipe_submit(mars_model_t *model) {
    if (!model) return MARS_ERR_INVALID_FILE;
    mars_model_ext_t *m = (mars_model_ext_t *)model;
    mars_pipe_t *pp = (mars_pipe_t *)m->pipe;
    if (!pp || m->batch != pp->batch) return MARS_ERR_INVALID_FILE; /* the batch was re-planned: open the pipe again */
    if (pp->inflight >= PIPE_DEPTH) return MARS_ERR_ALLOC_FAILED;    /* collect a result first (mars_hip_pipe_wait) */
    pipe_slot_t *sl = &pp->slot[pp->head];
    const size_t B = (size_t)m->batch;
    int rc = 0;
    /* ---- upload stream: the slot's device inputs were last read by the graph of its previous use */
    mhip_select_stream(2);
    if (sl->used) rc = mhip_stream_wait(2, sl->ev_graph);
    const int camera = pp->opts.camera_w > 0;
    for (uint32_t i = 0; i < model->header.num_inputs && !rc; i++) {
        mtensor_t *t = io_tensor(m, 0, (int)i);
        if (!t || !t->bytes) continue;
        if (i == 0 && camera) { /* RGB frames up; the front-end runs on the MAIN stream ahead of the graph (below): on this stream it stood between
                                 * batch k's copy and batch k + 1's, and the link idled while it ran (17.7 ms per batch = copy + kernel) */
            rc = mhip_h2d_async(sl->rgb_dev, sl->in_host[0], (size_t)pp->opts.camera_w * pp->opts.camera_h * 3 * B);
            continue;
        }
        rc = mhip_h2d_2d_async(sl->in_dev[i], t->stride, sl->in_host[i], t->bytes, t->bytes, B);
    }
    if (!rc) rc = mhip_event_record(sl->ev_up);
    /* ---- main stream: the graph on this slot's buffers */
    mhip_select_stream(0);
    if (!rc) rc = mhip_stream_wait(0, sl->ev_up);
    if (!rc && sl->used) { /* its outputs must have been consumed by the tail and the download of the previous use */
        rc = mhip_stream_wait(0, sl->ev_down);
        if (!rc && pp->opts.detect) rc = mhip_stream_wait(0, sl->ev_tail);
    }
    if (rc) { mhip_select_stream(0); return MARS_ERR_LAYER_FAILED; }
    for (uint32_t i = 0; i < model->header.num_inputs; i++) {
        mtensor_t *t = io_tensor(m, 0, (int)i);
        if (t) t->dev = sl->in_dev[i];
    }
    for (uint32_t i = 0; i < model->header.num_outputs; i++) {
        mtensor_t *t = io_tensor(m, 1, (int)i);
        if (t) t->dev = sl->out_dev[i];
    }
    if (camera) { /* letterbox / px - 128 into this slot's graph input (now the input tensor's device buffer), then the graph behind it */
        rc = mhip_event_record(sl->ev_pre0);
        if (!rc && mars_hip_preprocess_device(model, 0, sl->rgb_dev, pp->opts.camera_w, pp->opts.camera_h, 0, (int)B) != MARS_OK) rc = -1;
        if (!rc) rc = mhip_event_record(sl->ev_pre1);
        if (rc) return MARS_ERR_LAYER_FAILED;
    }
    m->tail_pending = 0; /* ordering is by the slot events here, not by the single-buffer hand-off of mars_hip_detect */
    mars_error_t e = mars_hip_run_device_async(model);
    if (e != MARS_OK) return e;
    if (pp->opts.download_outputs)
        for (uint32_t i = 0; i < model->header.num_outputs && !rc; i++) { /* padded pixel rows are packed on the device */
            mtensor_t *t = io_tensor(m, 1, (int)i);
            if (t && t->pix_stride)
                rc = mhip_unpad_rows(sl->out_dev[i], sl->out_dense[i], (t->bytes / (size_t)t->pix_c) * B, t->pix_c, t->pix_stride);
        }
    if (!rc) rc = mhip_event_record(sl->ev_graph);
    /* ---- auxiliary stream: decode + NMS of this batch, beside the next batch's graph */
    if (!rc && pp->opts.detect) {
        mhip_select_stream(1);
        rc = mhip_stream_wait(1, sl->ev_graph);
        if (!rc) rc = mars_detect_launch(m, pp->opts.det_outputs, pp->opts.n_det_outputs, pp->opts.nms_thresh, sl->det_dev, sl->cnt_dev);
        if (!rc) rc = mhip_event_record(sl->ev_tail);
    }
    /* ---- download stream */
    mhip_select_stream(3);
    if (!rc) rc = mhip_stream_wait(3, sl->ev_graph);
    if (!rc && pp->opts.detect) rc = mhip_stream_wait(3, sl->ev_tail);
    if (pp->opts.download_outputs)
        for (uint32_t i = 0; i < model->header.num_outputs && !rc; i++) {
            mtensor_t *t = io_tensor(m, 1, (int)i);
            if (!t || !t->bytes) continue;
            if (t->pix_stride) rc = mhip_d2h_async(sl->out_host[i], sl->out_dense[i], t->bytes * B);
            else rc = mhip_d2h_2d_async(sl->out_host[i], t->bytes, sl->out_dev[i], t->stride, t->bytes, B);
        }
    if (!rc && pp->opts.detect) {
        rc = mhip_d2h_async(sl->det_host, sl->det_dev, B * MARS_YOLO_MAX_DET * sizeof(mars_det_t));
        if (!rc) rc = mhip_d2h_async(sl->cnt_host, sl->cnt_dev, B * sizeof(int));
    }
    if (!rc) rc = mhip_event_record(sl->ev_down);
    mhip_select_stream(0);
    if (rc) return MARS_ERR_LAYER_FAILED;
    sl->used = 1;
    pp->head = (pp->head + 1) % PIPE_SLOTS;
    pp->inflight++;
    return MARS_OK;
}

/* camera mode: device time of the front-end kernel (letterbox + px - 128 of one batch) of the batch most recently handed out by
 * mars_hip_pipe_wait(), in milliseconds; < 0 = not available */
float mars_hip_pipe_camera_ms(mars_model_t *model) {
    if (!model) return -1.f;
    mars_pipe_t *pp = (mars_pipe_t *)((mars_model_ext_t *)model)->pipe;
    if (!pp || pp->opts.camera_w <= 0) return -1.f;
    const pipe_slot_t *sl = &pp->slot[(pp->tail + PIPE_SLOTS - 1) % PIPE_SLOTS];
    if (!sl->used || !sl->ev_pre0 || !sl->ev_pre1) return -1.f;
    return mhip_event_elapsed_ms(sl->ev_pre0, sl->ev_pre1);
}

mars_error_t mars_hip_pipe_wait(mars_model_t *model, const void **outputs, const mars_det_t **dets, const int **counts) {
    if (!model) return MARS_ERR_INVALID_FILE;
    mars_model_ext_t *m = (mars_model_ext_t *)model;
    mars_pipe_t *pp = (mars_pipe_t *)m->pipe;
    if (!pp || pp->inflight <= 0) return MARS_ERR_INVALID_FILE;
    pipe_slot_t *sl = &pp->slot[pp->tail];
    if (mhip_event_sync(sl->ev_down)) return MARS_ERR_LAYER_FAILED;
    if (outputs)
        for (uint32_t i = 0; i < model->header.num_outputs; i++) outputs[i] = sl->out_host[i];
    if (dets) *dets = sl->det_host;
    if (counts) *counts = sl->cnt_host;
    pp->tail = (pp->tail + 1) % PIPE_SLOTS;
    pp->inflight--;
    model->inference_count++;
    return MARS_OK;
}
