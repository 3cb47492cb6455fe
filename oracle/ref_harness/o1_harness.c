/*
 * o1_harness.c -- TEST SCAFFOLDING: oracle O1 = the reference's PUBLIC API
 * exactly as shipped (8 MiB arena, round-robin buffers), linked against the
 * reference's own mars_runtime.c.  Valid parity target only for graphs whose
 * tensors do not alias in that arena.
 * Call pattern follows reference src/mars/mars_test.c:33-148.
 */
#include <stdint.h>
#include <string.h>

#include "mars_runtime.h"

int nna_init(void);
void nna_deinit(void);
void ref_quiet(int on);

static size_t desc_bytes(const mars_tensor_t *t) {
    size_t n = 1;
    for (uint32_t i = 0; i < t->ndims && i < MARS_MAX_DIMS; i++) n *= (size_t)t->shape[i];
    size_t es = (t->dtype == MARS_DTYPE_FLOAT32 || t->dtype == MARS_DTYPE_INT32) ? 4
              : (t->dtype == MARS_DTYPE_INT16) ? 2 : 1;
    return n * es;
}

/* returns mars_error_t (<=0) or the number of output bytes produced (>0) */
long ref_o1_run_file(const char *path, const void *input, size_t in_bytes,
                     void *out, size_t out_cap) {
    ref_quiet(1);
    long rc;
    mars_model_t *m = NULL;
    nna_init(); /* re-zeroes the arena: runs are independent */
    mars_error_t e = mars_load_file(path, &m);
    if (e != MARS_OK) { rc = e; goto done; }
    mars_runtime_tensor_t *in = mars_get_input(m, 0);
    mars_runtime_tensor_t *o = mars_get_output(m, 0);
    if (!in || !o) { rc = MARS_ERR_INVALID_TENSOR; mars_free(m); goto done; }
    memcpy(in->vaddr, input, in_bytes);
    e = mars_run(m);
    if (e != MARS_OK) { rc = e; mars_free(m); goto done; }
    size_t ob = desc_bytes(&o->desc);
    if (ob > out_cap) ob = out_cap;
    memcpy(out, o->vaddr, ob);
    rc = (long)ob;
    mars_free(m);
done:
    ref_quiet(0);
    return rc;
}

/* alloc_size the reference's public API reports for input 0 / output 0 after mars_load_file
 * (reference mars_runtime.c:250-334); returns mars_error_t */
long ref_o1_io_alloc(const char *path, size_t out[2]) {
    ref_quiet(1);
    mars_model_t *m = NULL;
    nna_init();
    mars_error_t e = mars_load_file(path, &m);
    if (e == MARS_OK) {
        mars_runtime_tensor_t *in = mars_get_input(m, 0), *o = mars_get_output(m, 0);
        out[0] = in ? in->alloc_size : 0;
        out[1] = o ? o->alloc_size : 0;
        mars_free(m);
    }
    ref_quiet(0);
    return e;
}
