/*
 * nna_tensor.h -- small NHWC tensor handles (host side only).
 * Semantics of reference include/nna_tensor.h:29-141 / src/tensor.c:36-152:
 * create() owns nna_malloc'd storage, from_data() borrows, reshape() keeps
 * the element count, accessors tolerate NULL.
 */
#ifndef THINGINO_ACCEL_NNA_TENSOR_H
#define THINGINO_ACCEL_NNA_TENSOR_H

#include "nna_types.h"

#ifdef __cplusplus
extern "C" {
#endif

nna_tensor_t *nna_tensor_create(const nna_shape_t *shape, nna_dtype_t dtype,
                                nna_format_t format);
nna_tensor_t *nna_tensor_from_data(void *data, const nna_shape_t *shape,
                                   nna_dtype_t dtype, nna_format_t format);
void nna_tensor_destroy(nna_tensor_t *tensor);

void *nna_tensor_data(const nna_tensor_t *tensor);
const nna_shape_t *nna_tensor_shape(const nna_tensor_t *tensor);
nna_dtype_t nna_tensor_dtype(const nna_tensor_t *tensor);
size_t nna_tensor_numel(const nna_tensor_t *tensor);
size_t nna_tensor_bytes(const nna_tensor_t *tensor);
int nna_tensor_reshape(nna_tensor_t *tensor, const nna_shape_t *new_shape);

nna_shape_t nna_shape_make(int32_t n, int32_t h, int32_t w, int32_t c);

#ifdef __cplusplus
}
#endif
#endif
