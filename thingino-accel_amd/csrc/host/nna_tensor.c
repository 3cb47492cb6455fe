/*
 * nna_tensor.c -- NHWC tensor handles; host bookkeeping only.
 * Behaviour of reference src/tensor.c:36-152.
 */
#include <stdlib.h>

#include "nna_memory.h"
#include "nna_tensor.h"

static size_t elem_bytes(nna_dtype_t dt) {
    switch (dt) {
        case NNA_DTYPE_FLOAT32: case NNA_DTYPE_INT32: case NNA_DTYPE_UINT32: return 4;
        case NNA_DTYPE_FLOAT16: case NNA_DTYPE_INT16: case NNA_DTYPE_UINT16: return 2;
        case NNA_DTYPE_INT8: case NNA_DTYPE_UINT8: return 1;
        default: return 0;
    }
}

static size_t count(const nna_shape_t *s) {
    size_t n = 1;
    for (int i = 0; i < s->ndim; i++) n *= (size_t)s->dims[i];
    return n;
}

static nna_tensor_t *make(void *data, const nna_shape_t *shape, nna_dtype_t dtype, nna_format_t format, int own) {
    if (!shape || shape->ndim <= 0 || shape->ndim > 4) return NULL;
    nna_tensor_t *t = (nna_tensor_t *)malloc(sizeof(*t));
    if (!t) return NULL;
    t->shape = *shape;
    t->dtype = dtype;
    t->format = format;
    t->bytes = count(shape) * elem_bytes(dtype);
    t->owns_data = own;
    t->data = own ? nna_malloc(t->bytes) : data;
    if (!t->data) {
        free(t);
        return NULL;
    }
    return t;
}

nna_tensor_t *nna_tensor_create(const nna_shape_t *shape, nna_dtype_t dtype, nna_format_t format) {
    return make(NULL, shape, dtype, format, 1);
}

nna_tensor_t *nna_tensor_from_data(void *data, const nna_shape_t *shape, nna_dtype_t dtype, nna_format_t format) {
    if (!data) return NULL;
    return make(data, shape, dtype, format, 0);
}

void nna_tensor_destroy(nna_tensor_t *t) {
    if (!t) return;
    if (t->owns_data && t->data) nna_free(t->data);
    free(t);
}

void *nna_tensor_data(const nna_tensor_t *t) { return t ? t->data : NULL; }
const nna_shape_t *nna_tensor_shape(const nna_tensor_t *t) { return t ? &t->shape : NULL; }
nna_dtype_t nna_tensor_dtype(const nna_tensor_t *t) { return t ? t->dtype : NNA_DTYPE_FLOAT32; }
size_t nna_tensor_numel(const nna_tensor_t *t) { return t ? count(&t->shape) : 0; }
size_t nna_tensor_bytes(const nna_tensor_t *t) { return t ? t->bytes : 0; }

int nna_tensor_reshape(nna_tensor_t *t, const nna_shape_t *ns) {
    if (!t || !ns) return NNA_ERROR_INVALID;
    if (count(&t->shape) != count(ns)) return NNA_ERROR_INVALID;
    t->shape = *ns;
    return NNA_SUCCESS;
}

nna_shape_t nna_shape_make(int32_t n, int32_t h, int32_t w, int32_t c) {
    nna_shape_t s = {{n, h, w, c}, 4};
    return s;
}
