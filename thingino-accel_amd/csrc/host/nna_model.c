/*
 * nna_model.c -- the reference's model-handle API (include/nna_model.h:45-116, src/model.c:168-590) on top of
 * the `.mars` executor: lets callers written against nna_model_load / get_input / run / get_output
 * (examples/test_inference.c:142-238) drive the GPU path.  SURVEY.md 8(f) rank 3.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mars_hip.h"
#include "mars_internal.h"
#include "mars_runtime.h"
#include "nna.h"
#include "nna_model.h"

struct nna_model {
    mars_model_t *mars;
    size_t file_size;
    int profiling;
    nna_tensor_t **in, **out; /* handles onto the staging buffers, made on first request (src/model.c:335-411) */
};

static nna_dtype_t dtype_of(uint32_t mars_dtype) {
    switch (mars_dtype) {
        case MARS_DTYPE_FLOAT32: return NNA_DTYPE_FLOAT32;
        case MARS_DTYPE_INT32: return NNA_DTYPE_INT32;
        case MARS_DTYPE_INT16: return NNA_DTYPE_INT16;
        case MARS_DTYPE_UINT8: return NNA_DTYPE_UINT8;
        default: return NNA_DTYPE_INT8;
    }
}

static nna_model_t *wrap(mars_model_t *mm, size_t file_size, const nna_model_options_t *opt) {
    nna_model_t *m = (nna_model_t *)calloc(1, sizeof(*m));
    if (!m) { mars_free(mm); return NULL; }
    m->mars = mm;
    m->file_size = file_size;
    m->profiling = opt && opt->enable_profiling;
    m->in = (nna_tensor_t **)calloc(mm->header.num_inputs ? mm->header.num_inputs : 1, sizeof(*m->in));
    m->out = (nna_tensor_t **)calloc(mm->header.num_outputs ? mm->header.num_outputs : 1, sizeof(*m->out));
    if (!m->in || !m->out) { nna_model_unload(m); return NULL; }
    if (m->profiling) mars_hip_set_profiling(mm, 1);
    return m;
}

nna_model_t *nna_model_load(const char *path, const nna_model_options_t *options) {
    if (!path || !nna_is_ready()) { /* src/model.c:169-176: needs nna_init() */
        fprintf(stderr, "nna_model_load: %s\n", path ? "NNA not initialized" : "NULL path");
        return NULL;
    }
    FILE *fh = fopen(path, "rb");
    if (!fh) { fprintf(stderr, "nna_model_load: cannot open %s\n", path); return NULL; }
    fseek(fh, 0, SEEK_END);
    const long n = ftell(fh);
    fclose(fh);
    mars_model_t *mm = NULL;
    if (mars_load_file(path, &mm) != MARS_OK || !mm) {
        fprintf(stderr, "nna_model_load: %s is not a loadable .mars graph (.mgk models are not supported here)\n", path);
        return NULL;
    }
    return wrap(mm, n > 0 ? (size_t)n : 0, options);
}

nna_model_t *nna_model_load_from_memory(const void *buffer, size_t size, const nna_model_options_t *options) {
    if (!buffer || size == 0 || !nna_is_ready()) return NULL;
    mars_model_t *mm = NULL;
    if (mars_load_memory(buffer, size, &mm) != MARS_OK || !mm) return NULL;
    return wrap(mm, size, options);
}

int nna_model_get_info(nna_model_t *model, nna_model_info_t *info) {
    if (!model || !info) return NNA_ERROR_INVALID;
    const mars_model_ext_t *x = (const mars_model_ext_t *)model->mars;
    info->num_inputs = model->mars->header.num_inputs;
    info->num_outputs = model->mars->header.num_outputs;
    info->num_layers = model->mars->header.num_layers;
    info->model_size = model->file_size;
    info->forward_mem_req = x->act_bytes;
    return NNA_SUCCESS;
}

/* bytes behind an I/O handle: every frame of the current batch, by shape (not the reference-style alloc_size) */
static size_t io_bytes(mars_model_t *m, uint32_t tid) {
    return mars_hip_tensor_frame_bytes(m, (int)tid) * (size_t)mars_hip_get_batch(m);
}

static nna_tensor_t *handle_for(const mars_runtime_tensor_t *rt, size_t bytes) {
    if (!rt || !rt->vaddr) return NULL;
    nna_shape_t shape;
    shape.ndim = (int32_t)(rt->desc.ndims > 4 ? 4 : rt->desc.ndims);
    for (int i = 0; i < 4; i++) shape.dims[i] = i < shape.ndim ? rt->desc.shape[i] : 1;
    /* the tensor borrows the staging buffer; its byte count is the buffer's (a batch > 1 makes it longer than the
     * shape says: frames are frame-major) */
    nna_tensor_t *t = nna_tensor_from_data(rt->vaddr, &shape, dtype_of(rt->desc.dtype), NNA_FORMAT_NHWC);
    if (t) t->bytes = bytes;
    return t;
}

nna_tensor_t *nna_model_get_input(nna_model_t *model, uint32_t index) {
    if (!model || index >= model->mars->header.num_inputs) return NULL;
    if (!model->in[index]) model->in[index] = handle_for(mars_get_input(model->mars, (int)index), io_bytes(model->mars, model->mars->header.input_tensor_ids[index]));
    return model->in[index];
}

const nna_tensor_t *nna_model_get_output(nna_model_t *model, uint32_t index) {
    if (!model || index >= model->mars->header.num_outputs) return NULL;
    if (!model->out[index]) model->out[index] = handle_for(mars_get_output(model->mars, (int)index), io_bytes(model->mars, model->mars->header.output_tensor_ids[index]));
    return model->out[index];
}

nna_tensor_t *nna_model_get_input_by_name(nna_model_t *model, const char *name) {
    if (!model || !name) return NULL;
    for (uint32_t i = 0; i < model->mars->header.num_inputs; i++) {
        const mars_runtime_tensor_t *rt = mars_get_input(model->mars, (int)i);
        if (rt && !strncmp(rt->desc.name, name, sizeof(rt->desc.name))) return nna_model_get_input(model, i);
    }
    return NULL;
}

const nna_tensor_t *nna_model_get_output_by_name(nna_model_t *model, const char *name) {
    if (!model || !name) return NULL;
    for (uint32_t i = 0; i < model->mars->header.num_outputs; i++) {
        const mars_runtime_tensor_t *rt = mars_get_output(model->mars, (int)i);
        if (rt && !strncmp(rt->desc.name, name, sizeof(rt->desc.name))) return nna_model_get_output(model, i);
    }
    return NULL;
}

int nna_model_run(nna_model_t *model) {
    if (!model) return NNA_ERROR_INVALID;
    if (!nna_is_ready()) return NNA_ERROR_INIT;
    return mars_run(model->mars) == MARS_OK ? NNA_SUCCESS : NNA_ERROR_DEVICE;
}

void nna_model_unload(nna_model_t *model) {
    if (!model) return;
    if (model->mars) {
        if (model->profiling && model->mars->inference_count) {
            const int n = mars_hip_num_ops(model->mars);
            fprintf(stderr, "nna_model: %llu runs, %d launches per run; last run per launch:\n",
                    (unsigned long long)model->mars->inference_count, n);
            for (int i = 0; i < n; i++) {
                int layer = 0, kind = 0;
                double macs = 0, bytes = 0;
                float ms = 0;
                mars_hip_op_info(model->mars, i, &layer, &kind, &macs, &bytes, &ms);
                fprintf(stderr, "  launch %3d layer %3d kind %d  %8.1f us  %8.3f MMAC  %8.3f MB\n", i, layer, kind, ms * 1e3,
                        macs / 1e6, bytes / 1e6);
            }
        }
        for (uint32_t i = 0; model->in && i < model->mars->header.num_inputs; i++) nna_tensor_destroy(model->in[i]);
        for (uint32_t i = 0; model->out && i < model->mars->header.num_outputs; i++) nna_tensor_destroy(model->out[i]);
        mars_free(model->mars);
    }
    free(model->in);
    free(model->out);
    free(model);
}
