/*
 * orc_graph.c -- CPU restatement of the reference's graph walk.
 * TEST INFRASTRUCTURE (see orc.h).
 *
 * Loader:     reference src/mars/mars_runtime.c:126-200 (parse only)
 * Dispatch:   :1161-1224 (which layer kinds compute, which are no-ops, which fail)
 * Per layer:  :511-710 conv, :724-771 sigmoid, :774-905 mul/add, :908-960
 *             maxpool, :963-1000 concat, :1003-1044 upsample, :1047-1089 relu,
 *             :1092-1158 batchnorm
 * Memory:     oracle O2 -- one zero-initialised buffer per activation tensor
 *             (no round-robin arena), weight blob followed by zeros.
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "orc.h"

#define E_OK 0
#define E_MAGIC (-1)
#define E_VERSION (-2)
#define E_ALLOC (-3)
#define E_FILE (-4)
#define E_LAYER_FAILED (-6)
#define E_TENSOR (-7)
#define E_LAYER (-8)
#define NO_TENSOR 0xFFFFFFFFu

typedef struct {
    mars_tensor_t d;
    uint8_t *ptr;   /* activation buffer or pointer into blob */
    size_t bytes;   /* numel * elemsize by shape */
    size_t alloc;   /* bytes reachable behind ptr */
    int is_weight;
} otensor_t;

struct orc_graph {
    mars_header_t hdr;
    otensor_t *t;
    mars_layer_t *l;
    uint8_t *blob;
    size_t blob_alloc;
};

static size_t elem_size(uint32_t dtype) {
    switch (dtype) {
        case MARS_DTYPE_FLOAT32:
        case MARS_DTYPE_INT32: return 4;
        case MARS_DTYPE_INT16: return 2;
        default: return 1;
    }
}

static size_t shape_numel(const mars_tensor_t *d) {
    size_t n = 1;
    for (uint32_t i = 0; i < d->ndims && i < MARS_MAX_DIMS; i++) {
        if (d->shape[i] <= 0) return 0;
        n *= (size_t)d->shape[i];
    }
    return n;
}

orc_graph_t *orc_graph_open(const void *file, size_t size, size_t slack_mult, size_t slack_add,
                            int *err) {
    int dummy;
    if (!err) err = &dummy;
    *err = E_FILE;
    if (!file || size < sizeof(mars_header_t)) return NULL;
    const uint8_t *p = (const uint8_t *)file;
    orc_graph_t *g = (orc_graph_t *)calloc(1, sizeof(*g));
    if (!g) { *err = E_ALLOC; return NULL; }
    memcpy(&g->hdr, p, sizeof(g->hdr));
    if (g->hdr.magic != MARS_MAGIC) { *err = E_MAGIC; free(g); return NULL; }
    if (g->hdr.version_major != MARS_VERSION_MAJOR) { *err = E_VERSION; free(g); return NULL; }
    const size_t nt = g->hdr.num_tensors, nl = g->hdr.num_layers;
    const size_t table_end = sizeof(mars_header_t) + nt * sizeof(mars_tensor_t) + nl * sizeof(mars_layer_t);
    if (nt > (1u << 20) || nl > (1u << 20) || table_end > size) { free(g); return NULL; }
    g->t = (otensor_t *)calloc(nt ? nt : 1, sizeof(otensor_t));
    g->l = (mars_layer_t *)calloc(nl ? nl : 1, sizeof(mars_layer_t));
    const uint8_t *q = p + sizeof(mars_header_t);
    for (size_t i = 0; i < nt; i++, q += sizeof(mars_tensor_t)) memcpy(&g->t[i].d, q, sizeof(mars_tensor_t));
    for (size_t i = 0; i < nl; i++, q += sizeof(mars_layer_t)) memcpy(&g->l[i], q, sizeof(mars_layer_t));

    size_t woff = g->hdr.weights_offset > size ? size : (size_t)g->hdr.weights_offset;
    size_t wsz = g->hdr.weights_size;
    if (wsz > size - woff) wsz = size - woff;
    g->blob_alloc = (size_t)g->hdr.weights_size + (1u << 20);
    g->blob = (uint8_t *)calloc(1, g->blob_alloc);
    memcpy(g->blob, p + woff, wsz);

    for (size_t i = 0; i < nt; i++) {
        otensor_t *t = &g->t[i];
        t->bytes = shape_numel(&t->d) * elem_size(t->d.dtype);
        if (t->d.data_size > 0) {
            t->is_weight = 1;
            if (t->d.data_offset >= g->blob_alloc) { t->ptr = NULL; t->alloc = 0; continue; }
            t->ptr = g->blob + t->d.data_offset;
            t->alloc = g->blob_alloc - (size_t)t->d.data_offset;
        } else {
            t->alloc = t->bytes * slack_mult + slack_add;
            if (t->alloc < 64) t->alloc = 64;
            t->ptr = (uint8_t *)calloc(1, t->alloc);
        }
    }
    *err = E_OK;
    return g;
}

void orc_graph_close(orc_graph_t *g) {
    if (!g) return;
    for (uint32_t i = 0; i < g->hdr.num_tensors; i++)
        if (!g->t[i].is_weight) free(g->t[i].ptr);
    free(g->blob);
    free(g->t);
    free(g->l);
    free(g);
}

void orc_graph_zero_activations(orc_graph_t *g) {
    for (uint32_t i = 0; i < g->hdr.num_tensors; i++)
        if (!g->t[i].is_weight && g->t[i].ptr) memset(g->t[i].ptr, 0, g->t[i].alloc);
}

int orc_graph_num_tensors(const orc_graph_t *g) { return (int)g->hdr.num_tensors; }
int orc_graph_num_layers(const orc_graph_t *g) { return (int)g->hdr.num_layers; }

int orc_graph_input_id(const orc_graph_t *g, int i) {
    if (i < 0 || (uint32_t)i >= g->hdr.num_inputs || i >= 4) return -1;
    uint32_t id = g->hdr.input_tensor_ids[i];
    return id < g->hdr.num_tensors ? (int)id : -1; /* index, as mars_get_input does (:399-401) */
}

int orc_graph_output_id(const orc_graph_t *g, int i) {
    if (i < 0 || (uint32_t)i >= g->hdr.num_outputs || i >= 4) return -1;
    uint32_t id = g->hdr.output_tensor_ids[i];
    return id < g->hdr.num_tensors ? (int)id : -1;
}

size_t orc_graph_tensor_bytes(const orc_graph_t *g, int ti) {
    if (ti < 0 || (uint32_t)ti >= g->hdr.num_tensors) return 0;
    return g->t[ti].bytes;
}

void *orc_graph_tensor(orc_graph_t *g, int ti, size_t *alloc) {
    if (ti < 0 || (uint32_t)ti >= g->hdr.num_tensors) return NULL;
    if (alloc) *alloc = g->t[ti].alloc;
    return g->t[ti].ptr;
}

int orc_graph_set_input(orc_graph_t *g, int input_index, const void *data, size_t bytes) {
    int ti = orc_graph_input_id(g, input_index);
    if (ti < 0 || bytes > g->t[ti].alloc) return -1;
    memcpy(g->t[ti].ptr, data, bytes);
    return 0;
}

/* first tensor whose desc.id matches (the executor searches, it does not index) */
static otensor_t *by_id(orc_graph_t *g, uint32_t id) {
    if (id == NO_TENSOR) return NULL;
    for (uint32_t i = 0; i < g->hdr.num_tensors; i++)
        if (g->t[i].d.id == id) return &g->t[i];
    return NULL;
}

static size_t numel_of(const otensor_t *t) { return shape_numel(&t->d); }

static int run_conv(orc_graph_t *g, const mars_layer_t *L) {
    const mars_conv_params_t *cp = &L->params.conv;
    otensor_t *in = by_id(g, L->input_tensor_ids[0]);
    otensor_t *out = by_id(g, L->output_tensor_ids[0]);
    otensor_t *w = by_id(g, cp->weight_tensor_id);
    if (!in || !in->ptr || !out || !out->ptr || !w || !w->ptr) return E_TENSOR;
    otensor_t *b = by_id(g, cp->bias_tensor_id);

    orc_conv_geom_t q;
    const int in_nhwc = in->d.format == MARS_FORMAT_NHWC;
    const int out_nhwc = out->d.format == MARS_FORMAT_NHWC;
    if (in_nhwc) { q.in_h = in->d.shape[1]; q.in_w = in->d.shape[2]; q.in_c = in->d.shape[3]; }
    else         { q.in_c = in->d.shape[1]; q.in_h = in->d.shape[2]; q.in_w = in->d.shape[3]; }
    if (out_nhwc) { q.out_h = out->d.shape[1]; q.out_w = out->d.shape[2]; q.out_c = out->d.shape[3]; }
    else          { q.out_c = out->d.shape[1]; q.out_h = out->d.shape[2]; q.out_w = out->d.shape[3]; }
    q.kh = (int)cp->kernel_h; q.kw = (int)cp->kernel_w;
    q.stride_h = (int)cp->stride_h; q.stride_w = (int)cp->stride_w;
    q.pad_top = q.pad_left = 0;
    if (cp->padding == MARS_PAD_SAME) { /* only SAME pads; EXPLICIT/VALID run unpadded (:592-598) */
        int32_t ph = (int32_t)((uint32_t)(q.out_h - 1) * cp->stride_h + cp->kernel_h - (uint32_t)q.in_h);
        int32_t pw = (int32_t)((uint32_t)(q.out_w - 1) * cp->stride_w + cp->kernel_w - (uint32_t)q.in_w);
        q.pad_top = ph / 2;
        q.pad_left = pw / 2;
    }
    if (q.in_h < 0 || q.in_w < 0 || q.in_c < 0 || q.out_h < 0 || q.out_w < 0 || q.out_c < 0) return E_TENSOR;

    if (in->d.dtype == MARS_DTYPE_FLOAT32) {
        orc_conv_f32_nchw((const float *)in->ptr, (const float *)w->ptr,
                          b && b->ptr ? (const float *)b->ptr : NULL, (float *)out->ptr, &q);
    } else if (in_nhwc) {
        orc_conv_i8_nhwc((const int8_t *)in->ptr, (const int8_t *)w->ptr,
                         b && b->ptr ? (const int32_t *)b->ptr : NULL, (int8_t *)out->ptr, &q,
                         in->d.scale, w->d.scale, out->d.scale);
    } else {
        orc_conv_i8_nchw((const int8_t *)in->ptr, (const int8_t *)w->ptr,
                         b && b->ptr ? (const int32_t *)b->ptr : NULL, (int8_t *)out->ptr, &q,
                         in->d.scale, w->d.scale, out->d.scale);
    }
    if (cp->activation == MARS_ACT_RELU)
        orc_relu_bytes((int8_t *)out->ptr, (size_t)q.out_h * q.out_w * q.out_c);
    return E_OK;
}

static int run_layer(orc_graph_t *g, const mars_layer_t *L) {
    switch (L->type) {
        case MARS_LAYER_CONV2D: return run_conv(g, L);

        case MARS_LAYER_SIGMOID: {
            otensor_t *a = by_id(g, L->input_tensor_ids[0]), *o = by_id(g, L->output_tensor_ids[0]);
            if (!a || !o || !a->ptr || !o->ptr) return E_TENSOR;
            size_t n = numel_of(a);
            if (a->d.dtype == MARS_DTYPE_FLOAT32) orc_sigmoid_f32((const float *)a->ptr, (float *)o->ptr, n);
            else orc_sigmoid_i8((const int8_t *)a->ptr, (int8_t *)o->ptr, n, a->d.scale, o->d.scale);
            return E_OK;
        }
        case MARS_LAYER_MUL:
        case MARS_LAYER_ADD: {
            otensor_t *a = by_id(g, L->input_tensor_ids[0]), *b = by_id(g, L->input_tensor_ids[1]);
            otensor_t *o = by_id(g, L->output_tensor_ids[0]);
            if (!a || !b || !o || !a->ptr || !b->ptr || !o->ptr) return E_TENSOR;
            size_t n = numel_of(a); /* extent comes from the FIRST operand only */
            int mul = L->type == MARS_LAYER_MUL;
            if (a->d.dtype == MARS_DTYPE_FLOAT32)
                orc_binary_f32(mul, (const float *)a->ptr, (const float *)b->ptr, (float *)o->ptr, n);
            else
                orc_binary_i8(mul, (const int8_t *)a->ptr, (const int8_t *)b->ptr, (int8_t *)o->ptr, n,
                              a->d.scale, b->d.scale, o->d.scale);
            return E_OK;
        }
        case MARS_LAYER_RELU:
        case MARS_LAYER_RELU6:      /* executed as plain ReLU, no upper clamp */
        case MARS_LAYER_LEAKY_RELU: {
            otensor_t *a = by_id(g, L->input_tensor_ids[0]), *o = by_id(g, L->output_tensor_ids[0]);
            if (!a || !o || !a->ptr || !o->ptr) return E_TENSOR;
            size_t n = numel_of(a);
            int leaky = L->type == MARS_LAYER_LEAKY_RELU;
            if (a->d.dtype == MARS_DTYPE_FLOAT32) orc_relu_f32((const float *)a->ptr, (float *)o->ptr, n, leaky);
            else orc_relu_i8((const int8_t *)a->ptr, (int8_t *)o->ptr, n, leaky);
            return E_OK;
        }
        case MARS_LAYER_MAXPOOL: {
            const mars_pool_params_t *pp = &L->params.pool;
            otensor_t *a = by_id(g, L->input_tensor_ids[0]), *o = by_id(g, L->output_tensor_ids[0]);
            if (!a || !o || !a->ptr || !o->ptr) return E_TENSOR;
            /* shape[1..3] read as H,W,C whatever the layout tag; int8 bytes whatever the dtype */
            orc_maxpool_i8((const int8_t *)a->ptr, (int8_t *)o->ptr, a->d.shape[1], a->d.shape[2],
                           a->d.shape[3], o->d.shape[1], o->d.shape[2], (int)pp->kernel_h,
                           (int)pp->kernel_w, (int)pp->stride_h, (int)pp->stride_w);
            return E_OK;
        }
        case MARS_LAYER_CONCAT: {
            otensor_t *o = by_id(g, L->output_tensor_ids[0]);
            if (!o || !o->ptr) return E_TENSOR;
            int off = 0;
            for (uint32_t k = 0; k < L->num_inputs && k < 4; k++) {
                otensor_t *a = by_id(g, L->input_tensor_ids[k]);
                if (!a || !a->ptr) continue; /* silently skipped (:980) */
                int in_c = a->d.shape[3];
                orc_concat_slice_i8((const int8_t *)a->ptr, (int8_t *)o->ptr, o->d.shape[1],
                                    o->d.shape[2], in_c, o->d.shape[3], off);
                off += in_c;
            }
            return E_OK;
        }
        case MARS_LAYER_UPSAMPLE: {
            const mars_upsample_params_t *up = &L->params.upsample;
            otensor_t *a = by_id(g, L->input_tensor_ids[0]), *o = by_id(g, L->output_tensor_ids[0]);
            if (!a || !o || !a->ptr || !o->ptr) return E_TENSOR;
            int in_h = a->d.shape[1], in_w = a->d.shape[2], ch = a->d.shape[3];
            int out_h = o->d.shape[1], out_w = o->d.shape[2];
            if ((up->scale_h == 0 && in_h == 0) || (up->scale_w == 0 && in_w == 0)) return E_TENSOR;
            int sh = up->scale_h > 0 ? (int)up->scale_h : out_h / in_h;
            int sw = up->scale_w > 0 ? (int)up->scale_w : out_w / in_w;
            if (sh == 0 || sw == 0) return E_TENSOR; /* the reference would divide by zero */
            orc_upsample_i8((const int8_t *)a->ptr, (int8_t *)o->ptr, in_h, in_w, ch, out_h, out_w, sh, sw);
            return E_OK;
        }
        case MARS_LAYER_BATCHNORM: {
            otensor_t *a = by_id(g, L->input_tensor_ids[0]), *o = by_id(g, L->output_tensor_ids[0]);
            otensor_t *s = by_id(g, L->input_tensor_ids[1]), *b = by_id(g, L->input_tensor_ids[2]);
            if (!a || !o || !a->ptr || !o->ptr) return E_TENSOR;
            int n = a->d.shape[0] > 0 ? a->d.shape[0] : 1, c = a->d.shape[1] > 0 ? a->d.shape[1] : 1;
            int h = a->d.shape[2] > 0 ? a->d.shape[2] : 1, w = a->d.shape[3] > 0 ? a->d.shape[3] : 1;
            const float *sp = s && s->ptr ? (const float *)s->ptr : NULL;
            const float *bp = b && b->ptr ? (const float *)b->ptr : NULL;
            if (a->d.dtype == MARS_DTYPE_FLOAT32)
                orc_batchnorm_f32((const float *)a->ptr, (float *)o->ptr, n, c, h, w, sp, bp);
            else
                orc_batchnorm_i8((const int8_t *)a->ptr, (int8_t *)o->ptr, n, c, h, w, sp, bp,
                                 a->d.scale, o->d.scale);
            return E_OK;
        }
        /* accepted and ignored: the output buffer keeps whatever it held (:1168-1213) */
        case MARS_LAYER_DEPTHWISE_CONV2D:
        case MARS_LAYER_AVGPOOL:
        case MARS_LAYER_SILU:
        case MARS_LAYER_RESHAPE:
        case MARS_LAYER_TRANSPOSE:
        case MARS_LAYER_SOFTMAX: return E_OK;
        default: return E_LAYER; /* GLOBAL_AVGPOOL, FC, unknown (:1218-1220) */
    }
}

int orc_graph_run_range(orc_graph_t *g, int first, int last) {
    for (int i = first; i < last && (uint32_t)i < g->hdr.num_layers; i++) {
        int rc = run_layer(g, &g->l[i]);
        if (rc != E_OK) return rc;
    }
    return E_OK;
}

int orc_graph_run(orc_graph_t *g) { return orc_graph_run_range(g, 0, (int)g->hdr.num_layers); }

double orc_graph_conv_macs(const orc_graph_t *cg) {
    orc_graph_t *g = (orc_graph_t *)cg;
    double total = 0;
    for (uint32_t i = 0; i < g->hdr.num_layers; i++) {
        const mars_layer_t *L = &g->l[i];
        if (L->type != MARS_LAYER_CONV2D) continue;
        otensor_t *in = by_id(g, L->input_tensor_ids[0]), *out = by_id(g, L->output_tensor_ids[0]);
        if (!in || !out) continue;
        int nh = in->d.format == MARS_FORMAT_NHWC, oh = out->d.format == MARS_FORMAT_NHWC;
        double ic = nh ? in->d.shape[3] : in->d.shape[1];
        double oc = oh ? out->d.shape[3] : out->d.shape[1];
        double h = oh ? out->d.shape[1] : out->d.shape[2], w = oh ? out->d.shape[2] : out->d.shape[3];
        total += h * w * oc * ic * L->params.conv.kernel_h * L->params.conv.kernel_w;
    }
    return total;
}

/* ---------------------------------------------------- frames in parallel */
typedef struct {
    const void *file; size_t size;
    const uint8_t *in; size_t in_stride;
    uint8_t *out; size_t out_stride; int out_index;
    int nframes, nthreads, tid, rc;
} job_t;

static void *frame_worker(void *arg) {
    job_t *j = (job_t *)arg;
    int err = 0;
    orc_graph_t *g = orc_graph_open(j->file, j->size, 1, 4096, &err);
    if (!g) { j->rc = err ? err : E_FILE; return NULL; }
    int oi = orc_graph_output_id(g, j->out_index), ii = orc_graph_input_id(g, 0);
    if (oi < 0 || ii < 0) { j->rc = E_TENSOR; orc_graph_close(g); return NULL; }
    size_t ib = g->t[ii].bytes, ob = g->t[oi].bytes;
    for (int f = j->tid; f < j->nframes; f += j->nthreads) {
        if (f != j->tid) orc_graph_zero_activations(g);
        memcpy(g->t[ii].ptr, j->in + (size_t)f * j->in_stride, ib);
        j->rc = orc_graph_run(g);
        if (j->rc != E_OK) break;
        if (j->out) memcpy(j->out + (size_t)f * j->out_stride, g->t[oi].ptr, ob < j->out_stride ? ob : j->out_stride);
    }
    orc_graph_close(g);
    return NULL;
}

int orc_run_frames(const void *file, size_t size, const void *inputs, size_t in_stride,
                   void *outputs, size_t out_stride, int out_index, int nframes, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > nframes) nthreads = nframes > 0 ? nframes : 1;
    job_t *jobs = (job_t *)calloc((size_t)nthreads, sizeof(job_t));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (job_t){file, size, (const uint8_t *)inputs, in_stride, (uint8_t *)outputs,
                          out_stride, out_index, nframes, nthreads, t, 0};
        if (nthreads == 1) frame_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, frame_worker, &jobs[t]);
    }
    int rc = 0;
    for (int t = 0; t < nthreads; t++) {
        if (nthreads > 1) pthread_join(th[t], NULL);
        if (jobs[t].rc != 0 && rc == 0) rc = jobs[t].rc;
    }
    free(jobs);
    free(th);
    return rc;
}
