/*
 * nna_model.h -- model handle API of the reference (reference include/nna_model.h:17-116), here driving `.mars`
 * graphs on the GPU.  Same names, argument meaning and return conventions; the reference's implementation
 * (src/model.c) loads Ingenic `.mgk` ELF models through the Venus runtime, which is out of scope on this platform
 *  -- a `.mgk`/non-`.mars` file makes the loaders return NULL.
 *   - input/output tensors are handles onto the model's pinned host staging buffers (what mars_get_input()->vaddr
 *     points at): fill nna_tensor_data(input), nna_model_run(), read nna_tensor_data(output);
 *   - options->enable_profiling brackets every kernel launch with events; nna_model_unload() then prints the
 *     per-launch table to stderr;  use_file_mapping / forward_memory are accepted and ignored (weights and
 *     activations live in HBM).
 */
#ifndef THINGINO_ACCEL_NNA_MODEL_H
#define THINGINO_ACCEL_NNA_MODEL_H

#include <stddef.h>
#include <stdint.h>

#include "nna_tensor.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference :22-27 */
typedef struct {
    int use_file_mapping;
    int enable_profiling;
    void *forward_memory;
    size_t forward_mem_size;
} nna_model_options_t;

/* reference :30-36 */
typedef struct {
    uint32_t num_inputs;
    uint32_t num_outputs;
    uint32_t num_layers;
    size_t model_size;      /* bytes of the model file */
    size_t forward_mem_req; /* here: bytes of HBM holding the activations of one batch */
} nna_model_info_t;

nna_model_t *nna_model_load(const char *path, const nna_model_options_t *options);                    /* :45 */
nna_model_t *nna_model_load_from_memory(const void *buffer, size_t size,
                                        const nna_model_options_t *options);                          /* :55 */
int nna_model_get_info(nna_model_t *model, nna_model_info_t *info);                                   /* :65 */
nna_tensor_t *nna_model_get_input(nna_model_t *model, uint32_t index);                                /* :74 */
nna_tensor_t *nna_model_get_input_by_name(nna_model_t *model, const char *name);                      /* :83 */
const nna_tensor_t *nna_model_get_output(nna_model_t *model, uint32_t index);                         /* :92 */
const nna_tensor_t *nna_model_get_output_by_name(nna_model_t *model, const char *name);               /* :101 */
int nna_model_run(nna_model_t *model);                                                                /* :109 */
void nna_model_unload(nna_model_t *model);                                                            /* :116 */

#ifdef __cplusplus
}
#endif
#endif
