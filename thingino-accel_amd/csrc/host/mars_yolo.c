/*
 * mars_yolo.c -- host side of the detection tail (decode + NMS on the GPU).
 *
 * The reference keeps this in a demo program: src/mars/mars_yolo_test.c:80-104
 * (parse_output) and :107-130 (nms).  Here the host only tabulates the three
 * float functions of an int8 value the decode needs -- with the host libm, in
 * the reference's exact expression order -- and launches yolo_tail.hip.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mars_internal.h"
#include "nna.h"

/* lut[0..255] = (float)q*scale ; lut[256..511] = obj(q) ; lut[512..767] = 1 + expf(-val(q)) */
static void build_decode_lut(float scale, float *lut) {
    for (int q = -128; q < 128; q++) {
        float val = (float)q * scale;
        lut[q + 128] = val;
        lut[256 + q + 128] = 1.0f / (1.0f + expf((-(float)q) * scale)); /* :84 */
        lut[512 + q + 128] = 1.0f + expf(-val);                         /* :92 */
    }
}

/* value[q] strictly increasing over the whole table: the decode kernel may then take the class argmax on the int8 bytes
 * (first maximum of q == first maximum of value[q]); any NaN, a zero / negative / underflowing scale says no */
static int lut_increasing(const float *lut) {
    for (int i = 1; i < 256; i++)
        if (!(lut[i] > lut[i - 1])) return 0;
    return 1;
}

static int ensure_det_buffers(mars_model_ext_t *m, int frames) {
    if (m->det_cap >= frames && m->det_dev) return 0;
    if (m->det_dev) mhip_free(m->det_dev);
    if (m->det_counts_dev) mhip_free(m->det_counts_dev);
    m->det_dev = mhip_malloc((size_t)frames * MARS_YOLO_MAX_DET * sizeof(mars_det_t));
    m->det_counts_dev = (int *)mhip_malloc((size_t)frames * 2 * sizeof(int));
    if (!m->det_lut_dev) m->det_lut_dev = (float *)mhip_malloc(4 * 768 * sizeof(float));
    if (!m->det_dev || !m->det_counts_dev || !m->det_lut_dev) return -1;
    m->det_cap = frames;
    return 0;
}

/* decode tables of the listed outputs (they depend only on the output scales): built and uploaded once.  Synchronises. */
mars_error_t mars_detect_prepare(mars_model_ext_t *m, const int *output_indices, int n_outputs) {
    mars_model_t *model = &m->pub;
    if (!output_indices || n_outputs <= 0 || n_outputs > 4) return MARS_ERR_INVALID_TENSOR;
    if (!m->det_lut_dev) m->det_lut_dev = (float *)mhip_malloc(4 * 768 * sizeof(float));
    if (!m->det_lut_dev) return MARS_ERR_ALLOC_FAILED;
    int lut_stale = m->det_lut_n != n_outputs;
    for (int s = 0; s < n_outputs; s++) {
        mars_runtime_tensor_t *t = mars_get_output(model, output_indices[s]);
        if (!t || t->desc.dtype != MARS_DTYPE_INT8) return MARS_ERR_INVALID_TENSOR;
        if (memcmp(&m->det_lut_scale[s], &t->desc.scale, sizeof(float)) != 0) lut_stale = 1;
    }
    if (!lut_stale) return MARS_OK;
    float lut[4 * 768];
    for (int s = 0; s < n_outputs; s++) {
        float sc = mars_get_output(model, output_indices[s])->desc.scale;
        build_decode_lut(sc, lut + s * 768);
        m->det_lut_scale[s] = sc;
        m->det_lut_mono[s] = lut_increasing(lut + s * 768);
    }
    if (mhip_sync()) return MARS_ERR_LAYER_FAILED;
    if (mhip_h2d_async(m->det_lut_dev, lut, (size_t)n_outputs * 768 * sizeof(float)) || mhip_sync())
        return MARS_ERR_LAYER_FAILED; /* `lut` is on this stack frame */
    m->det_lut_n = n_outputs;
    return MARS_OK;
}

/* decode + NMS of the current batch on the CURRENT stream, from the output tensors' current device buffers into
 * dets_dev [batch][1000] / counts_dev [2 * batch].  Tables must be prepared.  0 or a launch error. */
int mars_detect_launch(mars_model_ext_t *m, const int *output_indices, int n_outputs, float nms_thresh, void *dets_dev,
                       int *counts_dev) {
    mars_model_t *model = &m->pub;
    mhip_detect_t p;
    memset(&p, 0, sizeof(p));
    for (int s = 0; s < n_outputs; s++) {
        if (output_indices[s] < 0 || (uint32_t)output_indices[s] >= model->header.num_outputs) return -1;
        uint32_t ti = model->header.output_tensor_ids[output_indices[s]];
        if (ti >= model->header.num_tensors || !m->mt[ti].dev) return -1;
        p.pred[s] = (const int8_t *)m->mt[ti].dev;
        p.stride[s] = m->mt[ti].stride;
        p.pix_c[s] = m->mt[ti].pix_c; p.pix_stride[s] = m->mt[ti].pix_stride;
        p.npred[s] = (int)(m->mt[ti].bytes / 85); /* rows of 85 int8: x,y,w,h,obj,80 classes */
        p.lut[s] = m->det_lut_dev + s * 768;
        p.mono[s] = m->det_lut_mono[s];
    }
    p.nseg = n_outputs;
    p.frames = m->batch;
    p.nms_thresh = nms_thresh;
    p.dets = dets_dev;
    p.counts = counts_dev;
    p.raw_counts = counts_dev + m->batch;
    p.do_nms = 1;
    return mhip_detect(&p);
}

mars_error_t mars_hip_detect_device(mars_model_t *model, const int *output_indices, int n_outputs, float nms_thresh) {
    if (!model || !output_indices || n_outputs <= 0 || n_outputs > 4) return MARS_ERR_INVALID_TENSOR;
    mars_model_ext_t *m = (mars_model_ext_t *)model;
    if (m->det_cap < m->batch || !m->det_dev) {
        if (mhip_sync()) return MARS_ERR_LAYER_FAILED; /* re-allocation: nothing may be in flight */
        if (ensure_det_buffers(m, m->batch)) return MARS_ERR_ALLOC_FAILED;
        m->det_lut_n = 0;
    }
    for (int s = 0; s < n_outputs; s++) {
        if (!mars_get_output(model, output_indices[s])) return MARS_ERR_INVALID_TENSOR;
        uint32_t ti = model->header.output_tensor_ids[output_indices[s]];
        if (!m->mt[ti].dev) return MARS_ERR_INVALID_TENSOR;
    }
    mars_error_t e = mars_detect_prepare(m, output_indices, n_outputs);
    if (e != MARS_OK) return e;
    /* The tail runs on the auxiliary stream: it starts when the graph launches enqueued so far
     * have finished, and the NEXT run's output-writing layers wait for it (mars_hip_run_device_async),
     * so decode/sort/NMS of batch k overlap the convolutions of batch k+1. */
    if (!m->ev_graph_done) m->ev_graph_done = mhip_event_create_sync();
    if (!m->ev_tail_done) m->ev_tail_done = mhip_event_create_sync();
    if (!m->ev_graph_done || !m->ev_tail_done) return MARS_ERR_ALLOC_FAILED;
    if (mhip_event_record(m->ev_graph_done)) return MARS_ERR_LAYER_FAILED;
    mhip_select_aux(1);
    int rc = mhip_stream_wait(1, m->ev_graph_done);
    if (!rc) rc = mars_detect_launch(m, output_indices, n_outputs, nms_thresh, m->det_dev, m->det_counts_dev);
    if (!rc) rc = mhip_event_record(m->ev_tail_done);
    mhip_select_aux(0);
    if (rc) return MARS_ERR_LAYER_FAILED;
    m->tail_pending = 1;
    return MARS_OK;
}

mars_error_t mars_hip_detect(mars_model_t *model, const int *output_indices, int n_outputs, float nms_thresh,
                             mars_det_t *dets, int *counts) {
    if (!dets || !counts) return MARS_ERR_INVALID_TENSOR;
    mars_error_t e = mars_hip_detect_device(model, output_indices, n_outputs, nms_thresh);
    if (e != MARS_OK) return e;
    mars_model_ext_t *m = (mars_model_ext_t *)model;
    if (mhip_sync()) return MARS_ERR_LAYER_FAILED; /* both streams: the tail ran on the auxiliary one */
    m->tail_pending = 0; /* ... and is complete: the next run needs no hand-off (and may replay its captured graph) */
    if (mhip_d2h_async(dets, m->det_dev, (size_t)m->batch * MARS_YOLO_MAX_DET * sizeof(mars_det_t)) ||
        mhip_d2h_async(counts, m->det_counts_dev, (size_t)m->batch * sizeof(int)) || mhip_sync())
        return MARS_ERR_LAYER_FAILED;
    return MARS_OK;
}

/* ---- host-pointer, single-frame forms with the reference's signatures */
static int need_device(const char *who) {
    if (nna_is_ready() || nna_init() == NNA_SUCCESS) return 0;
    fprintf(stderr, "%s: no MI355X device available\n", who);
    return -1;
}

int mars_yolo_parse_output(const int8_t *data, int npred, float scale, mars_det_t *dets, int maxd) {
    if (!data || !dets || npred <= 0 || maxd <= 0) return 0;
    if (maxd > MARS_YOLO_MAX_DET) maxd = MARS_YOLO_MAX_DET;
    if (need_device("mars_yolo_parse_output")) return -1;
    const size_t pb = (size_t)npred * 85;
    uint8_t *d = (uint8_t *)mhip_malloc(pb + 256 + 768 * 4 + MARS_YOLO_MAX_DET * sizeof(mars_det_t) + 64);
    if (!d) return -1;
    float lut[768];
    build_decode_lut(scale, lut);
    uint8_t *dl = d + ((pb + 255) & ~(size_t)255);
    uint8_t *dd = dl + 768 * 4;
    int *dc = (int *)(dd + MARS_YOLO_MAX_DET * sizeof(mars_det_t));
    mhip_detect_t p;
    memset(&p, 0, sizeof(p));
    p.pred[0] = (const int8_t *)d; p.stride[0] = 0; p.npred[0] = npred; p.lut[0] = (const float *)dl;
    p.mono[0] = lut_increasing(lut);
    p.nseg = 1; p.frames = 1; p.dets = dd; p.counts = dc; p.raw_counts = NULL; p.do_nms = 0;
    int n = -1;
    if (!mhip_h2d_async(d, data, pb) && !mhip_h2d_async(dl, lut, sizeof(lut)) && !mhip_detect(&p) &&
        !mhip_d2h_async(&n, dc, sizeof(int)) && !mhip_sync()) {
        if (n > maxd) n = maxd; /* the first maxd candidates in scan order */
        if (n > 0 && (mhip_d2h_async(dets, dd, (size_t)n * sizeof(mars_det_t)) || mhip_sync())) n = -1;
    } else {
        mhip_sync();
        n = -1;
    }
    mhip_free(d);
    return n;
}

int mars_yolo_nms(mars_det_t *dets, int n, float thresh) {
    if (!dets || n <= 0) return 0;
    if (n > MARS_YOLO_MAX_DET) return -1;
    if (need_device("mars_yolo_nms")) return -1;
    uint8_t *d = (uint8_t *)mhip_malloc(MARS_YOLO_MAX_DET * sizeof(mars_det_t) + 64);
    if (!d) return -1;
    int *dc = (int *)(d + MARS_YOLO_MAX_DET * sizeof(mars_det_t));
    int kept = -1;
    if (!mhip_h2d_async(d, dets, (size_t)n * sizeof(mars_det_t)) && !mhip_h2d_async(dc, &n, sizeof(int)) &&
        !mhip_nms_only(d, dc, n, thresh) && !mhip_d2h_async(&kept, dc, sizeof(int)) && !mhip_sync()) {
        if (kept > 0 && (mhip_d2h_async(dets, d, (size_t)kept * sizeof(mars_det_t)) || mhip_sync())) kept = -1;
    } else {
        mhip_sync();
        kept = -1;
    }
    mhip_free(d);
    return kept;
}
