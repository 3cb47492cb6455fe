/*
 * o2_harness.c -- TEST SCAFFOLDING: oracle O2 = the reference's own layer
 * functions on a PRIVATE, non-aliased arena.
 *
 * This TU textually includes reference src/mars/mars_runtime.c (from where it
 * lies; nothing is copied) to reach its `static execute_layer`.  The harness
 * builds the mars_model_t itself: the weight blob followed by zero slack, and
 * ONE zero-initialised buffer per activation tensor with generous zero slack,
 * so that every tensor behaves as if it lived alone in a zero-filled address
 * space.  Same reference arithmetic, race-free memory semantics
 * (SURVEY.md section 8c "O2").
 */
#include "mars_runtime.c" /* found via -I$(REF)/src/mars */

void ref_quiet(int on);

typedef struct {
    mars_model_t model;
    uint8_t *blob;
    size_t blob_alloc;
    size_t *bytes; /* shape bytes per tensor */
    size_t *alloc; /* allocation per tensor */
} o2_ctx_t;

static size_t o2_shape_bytes(const mars_tensor_t *t) {
    size_t n = 1;
    for (uint32_t i = 0; i < t->ndims && i < MARS_MAX_DIMS; i++) {
        if (t->shape[i] <= 0) return 0;
        n *= (size_t)t->shape[i];
    }
    size_t es = (t->dtype == MARS_DTYPE_FLOAT32 || t->dtype == MARS_DTYPE_INT32) ? 4
              : (t->dtype == MARS_DTYPE_INT16) ? 2 : 1;
    return n * es;
}

void *ref_o2_open(const void *file, size_t size, size_t slack_mult, size_t slack_add) {
    const uint8_t *p = (const uint8_t *)file;
    if (size < sizeof(mars_header_t)) return NULL;
    o2_ctx_t *c = (o2_ctx_t *)calloc(1, sizeof(*c));
    mars_model_t *m = &c->model;
    memcpy(&m->header, p, sizeof(mars_header_t));
    if (m->header.magic != MARS_MAGIC) { free(c); return NULL; }
    uint32_t nt = m->header.num_tensors, nl = m->header.num_layers;
    size_t need = sizeof(mars_header_t) + (size_t)nt * sizeof(mars_tensor_t) +
                  (size_t)nl * sizeof(mars_layer_t);
    if (need > size) { free(c); return NULL; }
    m->tensors = (mars_runtime_tensor_t *)calloc(nt ? nt : 1, sizeof(mars_runtime_tensor_t));
    m->layers = (mars_runtime_layer_t *)calloc(nl ? nl : 1, sizeof(mars_runtime_layer_t));
    c->bytes = (size_t *)calloc(nt ? nt : 1, sizeof(size_t));
    c->alloc = (size_t *)calloc(nt ? nt : 1, sizeof(size_t));
    const uint8_t *q = p + sizeof(mars_header_t);
    for (uint32_t i = 0; i < nt; i++, q += sizeof(mars_tensor_t))
        memcpy(&m->tensors[i].desc, q, sizeof(mars_tensor_t));
    for (uint32_t i = 0; i < nl; i++, q += sizeof(mars_layer_t))
        memcpy(&m->layers[i].desc, q, sizeof(mars_layer_t));

    size_t wsz = m->header.weights_size;
    size_t woff = m->header.weights_offset;
    if (woff > size) woff = size;
    if (wsz > size - woff) wsz = size - woff;
    c->blob_alloc = m->header.weights_size + (1u << 20);
    c->blob = (uint8_t *)calloc(1, c->blob_alloc);
    memcpy(c->blob, p + woff, wsz);
    m->ddr_base = c->blob;
    m->ddr_size = c->blob_alloc;
    m->weights = c->blob;
    m->weights_size = m->header.weights_size;

    for (uint32_t i = 0; i < nt; i++) {
        mars_runtime_tensor_t *rt = &m->tensors[i];
        if (rt->desc.data_size > 0) {
            rt->vaddr = c->blob + rt->desc.data_offset;
            rt->alloc_size = rt->desc.data_size;
            continue;
        }
        c->bytes[i] = o2_shape_bytes(&rt->desc);
        c->alloc[i] = c->bytes[i] * slack_mult + slack_add;
        if (c->alloc[i] < 64) c->alloc[i] = 64;
        rt->vaddr = calloc(1, c->alloc[i]);
        rt->alloc_size = c->alloc[i];
    }
    return c;
}

int ref_o2_num_tensors(void *h) { return (int)((o2_ctx_t *)h)->model.header.num_tensors; }
int ref_o2_num_layers(void *h) { return (int)((o2_ctx_t *)h)->model.header.num_layers; }

void *ref_o2_tensor(void *h, int idx, size_t *shape_bytes, size_t *alloc) {
    o2_ctx_t *c = (o2_ctx_t *)h;
    if (idx < 0 || (uint32_t)idx >= c->model.header.num_tensors) return NULL;
    if (shape_bytes) *shape_bytes = c->bytes[idx];
    if (alloc) *alloc = c->model.tensors[idx].alloc_size;
    return c->model.tensors[idx].vaddr;
}

int ref_o2_set_input(void *h, int input_index, const void *data, size_t bytes) {
    o2_ctx_t *c = (o2_ctx_t *)h;
    mars_runtime_tensor_t *t = mars_get_input(&c->model, input_index);
    if (!t || bytes > t->alloc_size) return -1;
    memcpy(t->vaddr, data, bytes);
    return 0;
}

/* run layers [first, last) in file order; returns first failing mars_error_t */
int ref_o2_run_range(void *h, int first, int last) {
    o2_ctx_t *c = (o2_ctx_t *)h;
    int rc = 0;
    ref_quiet(1);
    for (int i = first; i < last && (uint32_t)i < c->model.header.num_layers; i++) {
        rc = execute_layer(&c->model, &c->model.layers[i]);
        if (rc != MARS_OK) break;
    }
    ref_quiet(0);
    return rc;
}

int ref_o2_run(void *h) {
    return ref_o2_run_range(h, 0, (int)((o2_ctx_t *)h)->model.header.num_layers);
}

void ref_o2_close(void *h) {
    o2_ctx_t *c = (o2_ctx_t *)h;
    if (!c) return;
    for (uint32_t i = 0; i < c->model.header.num_tensors; i++)
        if (c->model.tensors[i].desc.data_size == 0) free(c->model.tensors[i].vaddr);
    free(c->blob);
    free(c->bytes);
    free(c->alloc);
    free(c->model.tensors);
    free(c->model.layers);
    free(c);
}

size_t ref_o2_tensor_byte_size(void *h, int idx) {
    o2_ctx_t *c = (o2_ctx_t *)h;
    return tensor_byte_size(&c->model.tensors[idx].desc); /* reference :80-124 */
}
