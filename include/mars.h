/*
 * mars.h -- on-disk layout of a .mars graph file (little-endian, packed).
 *
 * Drop-in for the reference header of the same name: every struct, enum value
 * and macro below has the byte layout / numeric value of
 *   reference include/mars.h:22-221
 * so that code written against the reference compiles and links unchanged.
 * The real sizes are header 76 B, tensor 124 B, layer 112 B, conv params 60 B
 * (the "64/64/128 byte" remarks in the reference header are stale); they are
 * pinned by the _Static_asserts at the bottom of this file.
 *
 * File = [mars_header_t][mars_tensor_t x num_tensors][mars_layer_t x num_layers]
 *        ... [weight blob at header.weights_offset, header.weights_size bytes]
 */
#ifndef MARS_H
#define MARS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MARS_MAGIC 0x5352414D /* 'M' 'A' 'R' 'S' */
#define MARS_VERSION_MAJOR 1
#define MARS_VERSION_MINOR 0

#define MARS_MAX_DIMS 6
#define MARS_MAX_NAME_LEN 64
#define MARS_MAX_LAYERS 256
#define MARS_MAX_TENSORS 512

/* element types (reference mars.h:35-42) */
typedef enum {
    MARS_DTYPE_FLOAT32 = 0,
    MARS_DTYPE_INT32 = 1,
    MARS_DTYPE_INT16 = 2,
    MARS_DTYPE_INT8 = 3,
    MARS_DTYPE_UINT8 = 4,
    MARS_DTYPE_UINT4 = 5,
} mars_dtype_t;

/*
 * layout tags (reference mars.h:46-56).  The executor only ever asks
 * "is it NHWC (7)?"; every other value is walked as NCHW / OIHW
 * (reference mars_runtime.c:561-562).
 */
typedef enum {
    MARS_FORMAT_NCHW = 0,
    MARS_FORMAT_NDHWC32 = 1,
    MARS_FORMAT_HWIO = 2,
    MARS_FORMAT_NMHWSOIB2 = 3,
    MARS_FORMAT_NMC32 = 4,
    MARS_FORMAT_D1 = 5,
    MARS_FORMAT_OHWI = 6,
    MARS_FORMAT_NHWC = 7,
    MARS_FORMAT_OIHW = 8,
} mars_format_t;

/* layer kinds (reference mars.h:59-79) */
typedef enum {
    MARS_LAYER_CONV2D = 0,
    MARS_LAYER_DEPTHWISE_CONV2D = 1,
    MARS_LAYER_MAXPOOL = 2,
    MARS_LAYER_AVGPOOL = 3,
    MARS_LAYER_GLOBAL_AVGPOOL = 4,
    MARS_LAYER_RELU = 5,
    MARS_LAYER_RELU6 = 6,
    MARS_LAYER_LEAKY_RELU = 7,
    MARS_LAYER_SILU = 8,
    MARS_LAYER_SIGMOID = 9,
    MARS_LAYER_CONCAT = 10,
    MARS_LAYER_ADD = 11,
    MARS_LAYER_MUL = 12,
    MARS_LAYER_UPSAMPLE = 13,
    MARS_LAYER_RESHAPE = 14,
    MARS_LAYER_SOFTMAX = 15,
    MARS_LAYER_FC = 16,
    MARS_LAYER_TRANSPOSE = 17,
    MARS_LAYER_BATCHNORM = 18,
} mars_layer_type_t;

/* activation fused into conv/fc (reference mars.h:82-91) */
typedef enum {
    MARS_ACT_NONE = 0,
    MARS_ACT_RELU = 1,
    MARS_ACT_RELU6 = 2,
    MARS_ACT_LEAKY_RELU = 3,
    MARS_ACT_SILU = 4,
    MARS_ACT_SIGMOID = 5,
    MARS_ACT_TANH = 6,
    MARS_ACT_HARD_SWISH = 7,
} mars_activation_t;

/* padding mode (reference mars.h:94-98); only SAME is honoured at run time */
typedef enum {
    MARS_PAD_VALID = 0,
    MARS_PAD_SAME = 1,
    MARS_PAD_EXPLICIT = 2,
} mars_padding_t;

#define MARS_PACKED __attribute__((packed))

typedef struct MARS_PACKED {
    uint32_t magic;
    uint16_t version_major;
    uint16_t version_minor;
    uint32_t flags;
    uint32_t num_layers;
    uint32_t num_tensors;
    uint32_t num_inputs;
    uint32_t num_outputs;
    uint64_t weights_offset;
    uint64_t weights_size;
    uint32_t input_tensor_ids[4];
    uint32_t output_tensor_ids[4];
} mars_header_t;

typedef struct MARS_PACKED {
    uint32_t id;
    char name[MARS_MAX_NAME_LEN - 4];
    mars_dtype_t dtype;
    mars_format_t format;
    uint32_t ndims;
    int32_t shape[MARS_MAX_DIMS];
    uint64_t data_offset; /* byte offset into the weight blob */
    uint64_t data_size;   /* 0 => activation tensor, produced at run time */
    float scale;
    int32_t zero_point;   /* carried, never read by the executor */
} mars_tensor_t;

typedef struct MARS_PACKED {
    uint32_t kernel_h, kernel_w;
    uint32_t stride_h, stride_w;
    uint32_t dilation_h, dilation_w;
    mars_padding_t padding;
    uint32_t pad_top, pad_bottom, pad_left, pad_right;
    uint32_t groups;
    mars_activation_t activation;
    uint32_t weight_tensor_id;
    uint32_t bias_tensor_id; /* 0xFFFFFFFF: none */
} mars_conv_params_t;

typedef struct MARS_PACKED {
    uint32_t kernel_h, kernel_w;
    uint32_t stride_h, stride_w;
    mars_padding_t padding;
    uint32_t pad_top, pad_bottom, pad_left, pad_right;
} mars_pool_params_t;

typedef struct MARS_PACKED {
    float alpha;
} mars_act_params_t;

typedef struct MARS_PACKED {
    uint32_t axis;
    uint32_t num_inputs;
} mars_concat_params_t;

typedef struct MARS_PACKED {
    uint32_t scale_h, scale_w;
    uint32_t mode; /* 0 nearest, 1 bilinear (only nearest is executed) */
} mars_upsample_params_t;

typedef struct MARS_PACKED {
    int32_t new_shape[MARS_MAX_DIMS];
    uint32_t ndims;
} mars_reshape_params_t;

typedef struct MARS_PACKED {
    uint32_t weight_tensor_id;
    uint32_t bias_tensor_id;
    mars_activation_t activation;
} mars_fc_params_t;

typedef struct MARS_PACKED {
    uint32_t id;
    mars_layer_type_t type;
    uint32_t num_inputs;
    uint32_t num_outputs;
    uint32_t input_tensor_ids[4];
    uint32_t output_tensor_ids[4];
    union {
        mars_conv_params_t conv;
        mars_pool_params_t pool;
        mars_act_params_t act;
        mars_concat_params_t concat;
        mars_upsample_params_t upsample;
        mars_reshape_params_t reshape;
        mars_fc_params_t fc;
        uint8_t raw[64];
    } params;
} mars_layer_t;

#ifndef __cplusplus
_Static_assert(sizeof(mars_header_t) == 76, "mars_header_t ABI");
_Static_assert(sizeof(mars_tensor_t) == 124, "mars_tensor_t ABI");
_Static_assert(sizeof(mars_layer_t) == 112, "mars_layer_t ABI");
_Static_assert(sizeof(mars_conv_params_t) == 60, "mars_conv_params_t ABI");
#else
static_assert(sizeof(mars_header_t) == 76, "mars_header_t ABI");
static_assert(sizeof(mars_tensor_t) == 124, "mars_tensor_t ABI");
static_assert(sizeof(mars_layer_t) == 112, "mars_layer_t ABI");
static_assert(sizeof(mars_conv_params_t) == 60, "mars_conv_params_t ABI");
#endif

#ifdef __cplusplus
}
#endif
#endif /* MARS_H */
