/*
 * nna_types.h -- scalar codes and plain structs shared by the nna_* API.
 * Values and field order follow reference include/nna_types.h:18-63 so that
 * binaries built against the reference keep their meaning.
 */
#ifndef THINGINO_ACCEL_NNA_TYPES_H
#define THINGINO_ACCEL_NNA_TYPES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes (0 ok, negative = failure) */
#define NNA_SUCCESS 0
#define NNA_ERROR_INIT (-1)
#define NNA_ERROR_DEVICE (-2)
#define NNA_ERROR_MEMORY (-3)
#define NNA_ERROR_INVALID (-4)
#define NNA_ERROR_TIMEOUT (-5)

typedef enum {
    NNA_DTYPE_FLOAT32 = 0,
    NNA_DTYPE_FLOAT16 = 1,
    NNA_DTYPE_INT8 = 2,
    NNA_DTYPE_UINT8 = 3,
    NNA_DTYPE_INT16 = 4,
    NNA_DTYPE_UINT16 = 5,
    NNA_DTYPE_INT32 = 6,
    NNA_DTYPE_UINT32 = 7,
} nna_dtype_t;

typedef enum {
    NNA_FORMAT_NHWC = 1,
    NNA_FORMAT_NV12 = 5,
} nna_format_t;

typedef enum {
    NNA_MEM_DDR = 0,  /* here: pinned host memory, device-visible */
    NNA_MEM_ORAM = 1, /* here: accounting only (LDS is per-workgroup) */
} nna_mem_type_t;

typedef struct {
    int32_t dims[4]; /* N, H, W, C */
    int32_t ndim;
} nna_shape_t;

typedef struct {
    void *data;
    nna_shape_t shape;
    nna_dtype_t dtype;
    nna_format_t format;
    size_t bytes;
    int owns_data;
} nna_tensor_t;

/*
 * On MI355X: oram_* describe one CU's LDS (160 KiB), version is the gfx
 * target number (950).  vbase/pbase are 0: LDS has no host mapping.
 */
typedef struct {
    uint32_t oram_vbase;
    uint32_t oram_pbase;
    uint32_t oram_size;
    uint32_t version;
} nna_hw_info_t;

typedef struct nna_model nna_model_t;

#ifdef __cplusplus
}
#endif
#endif
