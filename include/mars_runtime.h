/*
 * mars_runtime.h -- the .mars graph executor API, MI355X implementation.
 *
 * Same names, argument meaning, struct field order and error codes as
 *   reference include/mars_runtime.h:19-138
 * (implementation replaced: reference src/mars/mars_runtime.c).  Callers of
 * the reference read mars_model_t / mars_runtime_tensor_t fields directly
 * (reference src/mars/mars_test.c:62-84, mars_yolo_test.c:154-189), so both
 * structs stay public and field-for-field identical; private GPU state lives
 * behind the public prefix and is never exposed.
 *
 * Meaning of the address fields on MI355X:
 *   vaddr  host-addressable pointer.  Graph inputs/outputs: pinned host
 *          staging (frame-major, `batch` frames).  Weights: into the host copy
 *          of the blob.  Internal activations: NULL (they live in HBM only;
 *          use mars_hip_read_tensor() from mars_hip.h to inspect them).
 *   paddr  the device (HBM) address of the same tensor.
 * Extensions for batching / device-resident I/O / the detection tail are in
 * mars_hip.h; nothing here changes for a single-frame caller.
 */
#ifndef MARS_RUNTIME_H
#define MARS_RUNTIME_H

#include "mars.h"
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference mars_runtime.h:19-29; strings by -err (mars_runtime.c:58-76) */
typedef enum {
    MARS_OK = 0,
    MARS_ERR_INVALID_MAGIC = -1,
    MARS_ERR_VERSION_MISMATCH = -2,
    MARS_ERR_ALLOC_FAILED = -3,
    MARS_ERR_INVALID_FILE = -4,
    MARS_ERR_NNA_INIT_FAILED = -5,
    MARS_ERR_LAYER_FAILED = -6,
    MARS_ERR_INVALID_TENSOR = -7,
    MARS_ERR_INVALID_LAYER = -8,
} mars_error_t;

typedef struct {
    mars_tensor_t desc;
    void *vaddr;
    void *paddr;
    size_t alloc_size;
    bool is_external;
} mars_runtime_tensor_t;

typedef struct {
    mars_layer_t desc;
    bool is_executed;
} mars_runtime_layer_t;

typedef struct {
    mars_header_t header;
    mars_runtime_tensor_t *tensors;
    mars_runtime_layer_t *layers;

    void *ddr_base;   /* host copy of the weight blob */
    void *ddr_paddr;  /* device base of the parameter arena */
    size_t ddr_size;
    void *oram_base;  /* unused on MI355X (LDS is not addressable from the host) */
    void *oram_paddr;
    size_t oram_size;

    void *weights;
    size_t weights_size;

    uint64_t total_inference_us; /* accumulated device time of mars_run calls */
    uint32_t inference_count;
} mars_model_t;

/* Parse + validate + upload + plan.  Requires a successful nna_init(). */
mars_error_t mars_load_file(const char *path, mars_model_t **model);
mars_error_t mars_load_memory(const void *data, size_t size, mars_model_t **model);
void mars_free(mars_model_t *model);

mars_runtime_tensor_t *mars_get_input(mars_model_t *model, int index);
mars_runtime_tensor_t *mars_get_output(mars_model_t *model, int index);
int mars_get_num_inputs(mars_model_t *model);
int mars_get_num_outputs(mars_model_t *model);

/* H2D of the input staging, every layer in file order on the GPU, D2H of the
 * outputs.  NULL model -> MARS_ERR_INVALID_FILE (reference mars_runtime.c:440). */
mars_error_t mars_run(mars_model_t *model);

const char *mars_get_error_string(mars_error_t err);
void mars_print_summary(mars_model_t *model);

#ifdef __cplusplus
}
#endif
#endif /* MARS_RUNTIME_H */
