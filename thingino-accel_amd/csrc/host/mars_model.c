/*
 * mars_model.c -- .mars loader of the MI355X executor: file validation, tensor / device state, batch allocation, accessors.
 * (The launch planner lives in mars_plan.c, the run paths in mars_run.c: one 2 200-line file until round 4.)
 *
 * Public behaviour: reference include/mars_runtime.h:79-138, implemented in
 * the reference by src/mars/mars_runtime.c.  What changes underneath:
 *   - the 8 MiB round-robin "DDR" arena (mars_runtime.c:205-334) becomes one
 *     HBM buffer per activation tensor and frame, zero-initialised, sized from
 *     the byte extents the layers actually touch (no aliasing between live
 *     tensors; the reference's in-place hazards cannot occur);
 *   - the per-call dispatch switch (mars_runtime.c:1161-1224) becomes a launch
 *     plan built once at load: tensor ids resolved, weights re-packed for the
 *     MFMA kernel, float scales folded on the host exactly as the C code folds
 *     them, 256-entry LUTs for every int8 transcendental;
 *   - a frame batch dimension (the reference has none).
 * Layer semantics, including the quirks, follow the reference line by line;
 * each planner below cites the lines it mirrors.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mars_internal.h"
#include "nna.h"


int mars_verbose(void) {
    static int v = -1;
    if (v < 0) v = getenv("MARS_VERBOSE") ? 1 : 0;
    return v;
}

void drop_graph(mars_model_ext_t *m) {
    if (m->graph_exec) {
        mhip_sync();
        mhip_graph_destroy(m->graph_exec);
    }
    m->graph_exec = NULL;
    m->ran_plain = 0;
}

/* ------------------------------------------------------------------ errors */
static const char *const k_err[] = {
    "OK", "Invalid magic number", "Version mismatch", "Memory allocation failed",
    "Invalid file format", "NNA initialization failed", "Layer execution failed",
    "Invalid tensor", "Invalid layer",
};

const char *mars_get_error_string(mars_error_t err) {
    int i = -(int)err;
    if (i >= 0 && i < (int)(sizeof(k_err) / sizeof(k_err[0]))) return k_err[i];
    return "Unknown error";
}

/* ------------------------------------------------------------------- load */
/* Ragged pixel rows of graph outputs (the 255-channel YOLO heads) are kept at a 16-byte-aligned pitch on the device:
 * the producing convolution then takes the aligned epilogue (16-byte stores straight from registers, every launch
 * form) instead of the LDS-staged copy-out with 8+4+2+1-byte row tails.  Only tensors nothing in the graph reads:
 * their readers are the detection tail (told the pitch) and the download / read entry points (2-D copies), so hosts
 * still see the reference's dense [H][W][C] bytes.  The pad channels carry zero weights; their bytes are never read. */
static void pad_output_rows(mars_model_ext_t *m) {
    for (uint32_t i = 0; i < m->pub.header.num_tensors; i++) m->mt[i].pix_c = m->mt[i].pix_stride = 0;
    if (getenv("MARS_HIP_NO_ROWPAD")) return;
    for (int i = 0; i < m->n_ops; i++) {
        mars_op_t *o = &m->ops[i];
        if (o->kind != OP_CONV_I8 || o->out_nchw || o->out_pix_stride || o->out_ch_off || o->add_t || o->t_out < 0 ||
            (o->out_c & 15) == 0 || (o->in_c & 15) != 0)
            continue;
        mtensor_t *t = &m->mt[o->t_out];
        const int P = (o->out_c + 15) & ~15;
        const size_t px = (size_t)o->out_h * o->out_w;
        if (!t->io_out || t->io_in || P > o->oc_pad || t->bytes != px * (size_t)o->out_c || (px * P) % 256 != 0) continue;
        int other = 0; /* any other op touching the tensor keeps it dense */
        for (int j = 0; j < m->n_ops && !other; j++) {
            if (j == i) continue;
            if (m->ops[j].t_out == o->t_out) other = 1;
            for (int k = 0; k < m->ops[j].n_in; k++)
                if (m->ops[j].t_in[k] == o->t_out) other = 1;
            for (int k = 0; k < m->ops[j].nseg; k++)
                if (m->ops[j].seg_t[k] == o->t_out) other = 1;
            if (m->ops[j].add_t - 1 == o->t_out) other = 1;
            for (int k = 0; k < m->ops[j].chain_n; k++)
                if (m->ops[j].chain_out[k] == o->t_out) other = 1;
        }
        if (other) continue;
        o->out_pix_stride = P;
        o->store_c = P;
        t->pix_c = o->out_c;
        t->pix_stride = P;
        if (px * P > t->extent) t->extent = px * P;
    }
}

static void free_device_state(mars_model_ext_t *m) {
    drop_graph(m);
    if (m->act_dev) mhip_free(m->act_dev);
    if (m->scratch_dev) mhip_free(m->scratch_dev);
    m->act_dev = m->scratch_dev = NULL;
    for (uint32_t i = 0; m->mt && i < m->pub.header.num_tensors; i++) {
        if (m->mt[i].host) mhip_host_free(m->mt[i].host);
        m->mt[i].host = NULL;
        if (m->mt[i].dense_dev) mhip_free(m->mt[i].dense_dev);
        m->mt[i].dense_dev = NULL;
        if (!m->mt[i].is_weight) m->mt[i].dev = NULL;
    }
    if (m->det_dev) mhip_free(m->det_dev);
    if (m->det_counts_dev) mhip_free(m->det_counts_dev);
    m->det_dev = NULL;
    m->det_counts_dev = NULL;
    m->det_cap = 0;
    m->tail_pending = 0;
}

static void free_ops(mars_model_ext_t *m) {
    drop_graph(m);
    for (int i = 0; i < m->n_ops; i++) {
        if (m->ops[i].ev0) mhip_event_destroy(m->ops[i].ev0);
        if (m->ops[i].ev1) mhip_event_destroy(m->ops[i].ev1);
    }
    free(m->ops);
    m->ops = NULL;
    m->n_ops = m->cap_ops = 0;
}

/* every loaded model (single-threaded contract, like the reference's globals: SURVEY 8b "Threading") */
static mars_model_ext_t *g_live_models = NULL;
mars_model_ext_t *mars_live_models(void) { return g_live_models; }

void mars_free(mars_model_t *model) {
    if (!model) return;
    mars_model_ext_t *m = (mars_model_ext_t *)model;
    for (mars_model_ext_t **pp = &g_live_models; *pp; pp = &(*pp)->live_next)
        if (*pp == m) { *pp = m->live_next; break; }
    if (m->pipe) mars_hip_pipe_close(model);
    if (mhip_ready()) mhip_sync();
    free_device_state(m);
    free_ops(m);
    if (m->arena_dev) mhip_free(m->arena_dev);
    if (m->det_lut_dev) mhip_free(m->det_lut_dev);
    if (m->ev_graph_done) mhip_event_destroy(m->ev_graph_done);
    if (m->ev_tail_done) mhip_event_destroy(m->ev_tail_done);
    if (m->ev_fork) mhip_event_destroy(m->ev_fork);
    for (int k = 0; k < 3; k++)
        if (m->ev_join[k]) mhip_event_destroy(m->ev_join[k]);
    for (int k = 0; k < 2; k++)
        for (int c = 0; c < 8; c++)
            if (m->ev_chunk[k][c]) mhip_event_destroy(m->ev_chunk[k][c]);
    free(m->arena_host);
    free(m->mt);
    free(m->pub.weights);
    free(m->pub.layers);
    free(m->pub.tensors);
    free(m);
}

mars_error_t build_plan(mars_model_ext_t *m) {
    const uint32_t nt = m->pub.header.num_tensors, nl = m->pub.header.num_layers;
    free_ops(m);
    m->arena_size = 0;
    if (m->arena_host) memset(m->arena_host, 0, m->arena_cap); /* re-plans must not see stale bytes */
    m->scratch_per_frame = 0;
    m->plan_err = MARS_OK;
    m->plan_f32_mode = mhip_conv_f32_mode(-1);
    m->blob_mirror_bytes = m->pub.weights_size;
    for (uint32_t i = 0; i < nt; i++) {
        m->mt[i].extent = m->mt[i].bytes;
        m->mt[i].pix_c = m->mt[i].pix_stride = 0;
        m->mt[i].needed = (m->mt[i].io_in || m->mt[i].io_out) ? 1 : 0;
    }
    /* slot 0 of the arena: mirror of the raw blob (element-wise layers may read weight
     * tensors directly); sized after planning, so reserve generously now */
    for (uint32_t i = 0; i < nl && m->plan_err == MARS_OK; i++) plan_layer(m, (int)i);
    if (m->plan_err != MARS_OK) return (mars_error_t)m->plan_err;
    if (m->fusion >= 1) {
        fuse_silu(m);
        fuse_lut(m); /* (after the SiLU chains: a lone activation layer behind a convolution) */
        fuse_silu_f32(m);
        nhwc_internal(m); /* (after the SiLU fold: its intermediates are gone; before Add folding / pairing: they then see NHWC convolutions) */
        fuse_add(m);
        fuse_add_f32(m);
        if (!m->no_vconcat) virtual_concat(m);
        elide_concat(m);
        trim_concat(m);
        fuse_pool_chains(m);
        pair_convs(m);
        if (m->fusion >= 2 && !m->no_bottleneck) fuse_bottleneck(m); /* opt-in (level 2); after pairing: a paired launch stays a pair */
        pad_output_rows(m);
        virtual_concat_q(m); /* (last: it splits a convolution in two launches over one output tensor) */
    }
    if (m->fusion < 1) nhwc_internal(m); /* (resets the tensors' flags) */
    f32_policy(m);
    zero_tail_f32(m); /* (before pairing: a pair shares its input, hence its limit; resets the flags at fusion level 0) */
    if (m->fusion >= 1) pair_convs_f32(m); /* (after f32_policy: a pair shares one kernel choice) */
    rec_pairs(m); /* (fusion >= 1 only; resets the tensors' record flags in any case) */
    virtual_concat_f32(m); /* (last: it replaces launches; shares rec_pairs' per-batch decision) */
    return (mars_error_t)m->plan_err;
}

mars_error_t upload_params(mars_model_ext_t *m) {
    /* final arena = [planned entries][blob mirror]; mirror appended last so its size is known */
    size_t mirror_off = arena_reserve(m, m->blob_mirror_bytes + 64);
    if (mirror_off == NO_OFF) return MARS_ERR_ALLOC_FAILED;
    if (!m->deferred) blob_read(m, 0, m->blob_mirror_bytes, m->arena_host + mirror_off);
    if (m->arena_dev) mhip_free(m->arena_dev);
    m->arena_dev = (uint8_t *)mhip_malloc(m->arena_size);
    if (!m->arena_dev) return MARS_ERR_ALLOC_FAILED;
    if (mhip_h2d_async(m->arena_dev, m->arena_host, m->arena_size) || mhip_sync()) return MARS_ERR_NNA_INIT_FAILED;
    for (uint32_t i = 0; i < m->pub.header.num_tensors; i++) {
        if (!m->mt[i].is_weight) continue;
        m->mt[i].dev = m->arena_dev + mirror_off + (size_t)m->pub.tensors[i].desc.data_offset;
        m->mt[i].stride = 0;
        m->pub.tensors[i].paddr = m->mt[i].dev;
    }
    m->pub.ddr_paddr = m->arena_dev;
    m->pub.ddr_size = m->arena_size;
    return MARS_OK;
}

mars_error_t alloc_batch(mars_model_ext_t *m, int n) {
    const uint32_t nt = m->pub.header.num_tensors;
    if (m->pipe) mars_hip_pipe_close(&m->pub); /* its slots were sized for the old batch */
    if (mhip_sync()) return MARS_ERR_LAYER_FAILED;
    /* tests exercise the two fall-backs below with a small batch: limits from the environment, read once per call */
    const char *env_v = getenv("MARS_HIP_VCONCAT_LIMIT"), *env_b = getenv("MARS_HIP_BOTTLENECK_LIMIT");
    const size_t vlim = env_v ? (size_t)strtoull(env_v, NULL, 0) : (size_t)0x7fffffffu, blim = env_b ? (size_t)strtoull(env_b, NULL, 0) : 0;
    /* the two batch-dependent fall-backs are decided afresh for every batch (they used to stick once a large
     * batch had switched them on): plan again with the fusions allowed, the checks below take them back if need be */
    /* (record-format pairs, rec_pairs: chosen per batch too -- plan again when a chosen pair would not fit this batch, or one left
     * alone at a larger batch may fit now) */
    const int rec_replan = (size_t)n > m->rec_max_frames || (m->rec_skipped && n < m->rec_frames);
    m->rec_frames = n;
    if (m->no_vconcat || m->no_bottleneck || rec_replan) {
        m->no_vconcat = m->no_bottleneck = 0;
        mars_error_t e = build_plan(m);
        if (e == MARS_OK) e = upload_params(m);
        if (e != MARS_OK) return e;
    }
    /* segmented (virtual concat) convolutions address their output with 32-bit buffer offsets: if this batch makes
     * an output tensor of one of them 2 GiB or more, plan again with materialised concats */
    if (!m->no_vconcat) /* deferred (descriptor-only) ranks take the same decision: it depends on shapes and batch only */
        for (int i = 0; i < m->n_ops; i++) {
            const mars_op_t *op = &m->ops[i];
            if (op->kind != OP_CONV_I8 || op->nseg < 2 || op->t_out < 0) continue;
            const mtensor_t *t = &m->mt[op->t_out];
            const size_t stride = ALIGN_UP(t->extent > t->bytes ? t->extent : t->bytes, 256);
            if (stride * (size_t)n > vlim) {
                m->no_vconcat = 1;
                mars_error_t e = build_plan(m);
                if (e == MARS_OK) e = upload_params(m);
                if (e != MARS_OK) return e;
                break;
            }
        }
    /* fused bottlenecks were accepted on geometry alone (one frame, no strides): the launcher's limits also depend on the
     * batch (32-bit output offsets, tile count).  The 1x1 has left the plan, so a launch that fails would have no
     * fallback: check every fused launch against THIS batch and plan again without the fusion if one does not fit */
    if (m->fusion >= 2 && !m->no_bottleneck)
        for (int i = 0; i < m->n_ops; i++) {
            const mars_op_t *op = &m->ops[i];
            if (op->kind != OP_CONV_I8 || !op->pre || op->t_out < 0) continue;
            const mtensor_t *t = &m->mt[op->t_out];
            mhip_conv_i8_t p;
            memset(&p, 0, sizeof(p));
            p.frames = n; p.in_c = op->in_c; p.in_h = op->in_h; p.in_w = op->in_w; p.out_h = op->out_h; p.out_w = op->out_w;
            p.out_c = op->store_c ? op->store_c : op->out_c; p.kh = op->kh; p.kw = op->kw; p.stride_h = op->sh; p.stride_w = op->sw;
            p.pad_top = op->pt; p.pad_left = op->pl; p.row_pad = op->row_pad; p.oc_pad = op->oc_pad; p.safe = op->safe;
            p.out_pix_stride = op->out_pix_stride; p.out_ch_off = op->out_ch_off;
            p.out_stride = ALIGN_UP(t->extent > t->bytes ? t->extent : t->bytes, 256);
            p.in_stride = op->t_in[0] >= 0 ? ALIGN_UP(m->mt[op->t_in[0]].extent > m->mt[op->t_in[0]].bytes ? m->mt[op->t_in[0]].extent : m->mt[op->t_in[0]].bytes, 256) : 0;
            p.pre_w = (const int8_t *)m; p.pre_bias = (const int32_t *)m; p.pre_lut2 = (const uint8_t *)m; p.lut2 = (const uint8_t *)m;
            p.lut = (const uint8_t *)m;
            if (!mhip_conv_i8_pre_ok(&p) || (blim && (size_t)n > blim)) {
                m->no_bottleneck = 1;
                mars_error_t e = build_plan(m);
                if (e == MARS_OK) e = upload_params(m);
                if (e != MARS_OK) return e;
                break;
            }
        }
    free_device_state(m);
    size_t per_frame = 0;
    for (uint32_t i = 0; i < nt; i++) {
        mtensor_t *t = &m->mt[i];
        if (t->is_weight || !t->needed) continue;
        t->stride = ALIGN_UP(t->extent > t->bytes ? t->extent : t->bytes, 256);
        if (t->stride == 0) t->stride = 256;
        per_frame += t->stride;
    }
    m->act_bytes = per_frame * (size_t)n;
    /* + 256: vector loads may read up to 15 bytes past the last element they use (the stem's unaligned 16-byte loads
     * of 12-byte pixel groups); the slack keeps that inside the allocation for the last tensor too */
    /* 256 in front: a convolution may read a tensor through a view that starts up to 256 bytes before it (virtual_concat_f32: those
     * results are overwritten, but the addresses must be mapped for the first tensor too) */
    m->act_dev = (uint8_t *)mhip_malloc(m->act_bytes + 512);
    if (!m->act_dev) return MARS_ERR_ALLOC_FAILED;
    if (mhip_memset_async(m->act_dev, 0, m->act_bytes + 256)) return MARS_ERR_ALLOC_FAILED;
    size_t off = 256;
    for (uint32_t i = 0; i < nt; i++) {
        mtensor_t *t = &m->mt[i];
        mars_runtime_tensor_t *rt = &m->pub.tensors[i];
        if (t->is_weight) continue;
        rt->vaddr = NULL; rt->paddr = NULL; rt->alloc_size = 0;
        if (!t->needed) continue;
        t->dev = m->act_dev + off;
        off += t->stride * (size_t)n;
        rt->paddr = t->dev;
        rt->alloc_size = t->stride * (size_t)n;
        if (t->io_in || t->io_out) {
            /* what a caller may fill / read through vaddr (mars_test.c:73-84): one frame = the reference's working
             * buffer size; a batch (extension) = n densely packed frames */
            const size_t hb = t->bytes * (size_t)n, ref = reference_buffer_size(m);
            size_t cap = hb > ref ? hb : ref;
            /* what the reference can ever report is bounded by its 8 MiB arena; anything larger comes from wrapped int
             * products of a hostile descriptor (NDHWC32 / NMHWSOIB2 shapes near INT_MAX) and is not honoured: a single-frame
             * load must not pin and clear gigabytes per I/O tensor */
            if (ref > ((size_t)8 << 20)) cap = hb;
            t->host = (uint8_t *)mhip_host_alloc(cap ? cap : 64);
            if (!t->host) return MARS_ERR_ALLOC_FAILED;
            memset(t->host, 0, cap);
            rt->vaddr = t->host;
            rt->alloc_size = n == 1 && cap >= ref ? ref : hb;
        }
        if (t->pix_stride) {
            t->dense_dev = (uint8_t *)mhip_malloc(t->bytes * (size_t)n);
            if (!t->dense_dev) return MARS_ERR_ALLOC_FAILED;
        }
    }
    if (m->scratch_per_frame) {
        m->scratch_dev = (uint8_t *)mhip_malloc(ALIGN_UP(m->scratch_per_frame, 256) * (size_t)n + 256); /* (+ 256: 16-byte loads at the last pixels) */
        if (!m->scratch_dev) return MARS_ERR_ALLOC_FAILED;
    }
    m->batch = n;
    m->frame0 = 0;
    m->run_frames = n;
    return mhip_sync() ? MARS_ERR_ALLOC_FAILED : MARS_OK;
}

/* the host-only part of a load: parse, validate, plan (weights packed into the host image of the parameter arena).  No device call. */
static mars_error_t load_host(const void *data, size_t size, unsigned flags, mars_model_ext_t **out_m) {
    if (!data || !out_m || size < sizeof(mars_header_t)) return MARS_ERR_INVALID_FILE;
    const uint8_t *p = (const uint8_t *)data;
    mars_header_t h;
    memcpy(&h, p, sizeof(h));
    if (h.magic != MARS_MAGIC) {
        VLOG("Invalid magic 0x%08x (expected 0x%08x)\n", h.magic, MARS_MAGIC);
        return MARS_ERR_INVALID_MAGIC;
    }
    if (h.version_major != MARS_VERSION_MAJOR) return MARS_ERR_VERSION_MISMATCH;
    /* superset of the reference: it trusts every count and offset (mars_runtime.c:172-200, 220) */
    if (h.num_tensors > 65536 || h.num_layers > 65536 || h.num_inputs > 4 || h.num_outputs > 4) return MARS_ERR_INVALID_FILE;
    const size_t tables = sizeof(h) + (size_t)h.num_tensors * sizeof(mars_tensor_t) + (size_t)h.num_layers * sizeof(mars_layer_t);
    if (tables > size) return MARS_ERR_INVALID_FILE;
    const int deferred = (flags & MARS_HIP_LOAD_DEFER_WEIGHTS) != 0;
    if (!deferred && h.weights_size > 0 && (h.weights_offset > size || h.weights_size > size - h.weights_offset))
        return MARS_ERR_INVALID_FILE;
    if (h.weights_size > ((uint64_t)1 << 36)) return MARS_ERR_INVALID_FILE;

    mars_model_ext_t *m = (mars_model_ext_t *)calloc(1, sizeof(*m));
    if (!m) return MARS_ERR_ALLOC_FAILED;
    m->pub.header = h;
    m->fusion = getenv("MARS_HIP_FUSION") ? atoi(getenv("MARS_HIP_FUSION")) : 1;
    m->deferred = deferred;
    m->pub.tensors = (mars_runtime_tensor_t *)calloc(h.num_tensors ? h.num_tensors : 1, sizeof(mars_runtime_tensor_t));
    m->pub.layers = (mars_runtime_layer_t *)calloc(h.num_layers ? h.num_layers : 1, sizeof(mars_runtime_layer_t));
    m->mt = (mtensor_t *)calloc(h.num_tensors ? h.num_tensors : 1, sizeof(mtensor_t));
    if (!m->pub.tensors || !m->pub.layers || !m->mt) { mars_free(&m->pub); return MARS_ERR_ALLOC_FAILED; }
    const uint8_t *q = p + sizeof(h);
    for (uint32_t i = 0; i < h.num_tensors; i++, q += sizeof(mars_tensor_t)) memcpy(&m->pub.tensors[i].desc, q, sizeof(mars_tensor_t));
    for (uint32_t i = 0; i < h.num_layers; i++, q += sizeof(mars_layer_t)) memcpy(&m->pub.layers[i].desc, q, sizeof(mars_layer_t));

    m->pub.weights_size = (size_t)h.weights_size;
    if (h.weights_size) {
        m->pub.weights = calloc(1, (size_t)h.weights_size);
        if (!m->pub.weights) { mars_free(&m->pub); return MARS_ERR_ALLOC_FAILED; }
        if (!deferred) memcpy(m->pub.weights, p + h.weights_offset, (size_t)h.weights_size);
    }
    m->pub.ddr_base = m->pub.weights;
    for (uint32_t i = 0; i < h.num_tensors; i++) {
        mars_runtime_tensor_t *rt = &m->pub.tensors[i];
        if (rt->desc.ndims > MARS_MAX_DIMS) { mars_free(&m->pub); return MARS_ERR_INVALID_FILE; }
        m->mt[i].bytes = shape_numel(&rt->desc) * elem_size(rt->desc.dtype);
        if (rt->desc.data_size > 0) {
            if (rt->desc.data_offset > h.weights_size) { mars_free(&m->pub); return MARS_ERR_INVALID_FILE; }
            m->mt[i].is_weight = 1;
            rt->vaddr = (uint8_t *)m->pub.weights + rt->desc.data_offset;
            rt->alloc_size = (size_t)rt->desc.data_size;
        }
    }
    for (uint32_t i = 0; i < h.num_inputs; i++) {
        uint32_t id = h.input_tensor_ids[i];
        if (id < h.num_tensors && !m->mt[id].is_weight) m->mt[id].io_in = (int)i + 1;
    }
    for (uint32_t i = 0; i < h.num_outputs; i++) {
        uint32_t id = h.output_tensor_ids[i];
        if (id < h.num_tensors && !m->mt[id].is_weight) m->mt[id].io_out = (int)i + 1;
    }

    mars_error_t err = build_plan(m);
    if (err != MARS_OK) { mars_free(&m->pub); return err; }
    VLOG("%u layers -> %d launches, parameter arena %zu bytes\n", h.num_layers, m->n_ops, m->arena_size);
    *out_m = m;
    return MARS_OK;
}

mars_error_t mars_hip_load_memory_ex(const void *data, size_t size, unsigned flags, mars_model_t **out_model) {
    if (!out_model) return MARS_ERR_INVALID_FILE;
    mars_model_ext_t *m = NULL;
    mars_error_t err = load_host(data, size, flags, &m);
    if (err != MARS_OK) return err;
    /* everything above is host-only; from here on a GPU is required.  No CPU fallback. */
    if (!nna_is_ready() || !mhip_ready()) {
        fprintf(stderr, "Mars: no initialised MI355X device (call nna_init first); refusing to load\n");
        mars_free(&m->pub);
        return MARS_ERR_NNA_INIT_FAILED;
    }
    err = upload_params(m);
    if (err == MARS_OK) err = alloc_batch(m, 1);
    if (err != MARS_OK) { mars_free(&m->pub); return err; }
    m->live_next = g_live_models;
    g_live_models = m;
    *out_model = &m->pub;
    return MARS_OK;
}

/* The plan of a file as text, WITHOUT a device (tests of the planner on the CPU: which layers fuse, which tensors are kept pixels x channels,
 * which K loops are cut ...): one line per launch -- "op I layer L KIND in T.. out T" + the flags that are set -- and one per tensor that is not
 * held as tagged.  Returns the length needed (the text is cut at cap); 0 on a file the loader rejects. */
size_t mars_hip_describe_plan(const void *data, size_t size, unsigned flags, char *out, size_t cap) {
    static const char *const kinds[] = {"conv_i8", "conv_f32", "relu_bytes", "lut_i8", "binary_i8", "sigmoid_f32", "binary_f32", "relu_f32", "bn",
                                        "maxpool", "concat_slice", "upsample", "upsample_q", "maxpool_q", "concat_q", "conv_f32_vhead", "fail"};
    mars_model_ext_t *m = NULL;
    if (load_host(data, size, flags, &m) != MARS_OK) return 0;
    size_t n = 0;
    char line[512];
#define EMIT() do { const size_t l = strlen(line); if (out && n < cap) memcpy(out + n, line, n + l <= cap ? l : cap - n); n += l; } while (0)
    for (int i = 0; i < m->n_ops; i++) {
        const mars_op_t *o = &m->ops[i];
        int k = snprintf(line, sizeof line, "op %d layer %d %s in", i, o->layer, o->kind >= 0 && o->kind <= OP_FAIL ? kinds[o->kind] : "?");
        for (int q = 0; q < o->n_in && q < 4; q++) k += snprintf(line + k, sizeof line - (size_t)k, " %d", o->t_in[q]);
        k += snprintf(line + k, sizeof line - (size_t)k, " out %d", o->t_out);
#define FLAG(cond, ...) if (cond) k += snprintf(line + k, sizeof line - (size_t)k, __VA_ARGS__)
        FLAG(o->kind == OP_CONV_I8 || o->kind == OP_CONV_F32 || o->kind == OP_CONV_F32_VHEAD, " k%dx%d s%d c%d->%d", o->kh, o->kw, o->sw, o->in_c, o->out_c);
        FLAG(o->nchw, " relayout"); FLAG(o->out_nchw, " planar_store"); FLAG(o->lut_off != NO_OFF, " lut"); FLAG(o->add_t, " add=%d", o->add_t - 1);
        FLAG(o->nseg, " seg=%d", o->nseg); FLAG(o->pair_next, " pair_next"); FLAG(o->pre, " pre"); FLAG(o->silu_f32, " silu");
        FLAG(o->k_limit, " k_limit=%d", o->k_limit); FLAG(o->in_rec, " in_rec=%d", o->in_rec); FLAG(o->out_rec, " out_rec");
        FLAG(o->vc_shift, " view=-%d", o->vc_shift); FLAG(o->vc_n, " vcat=%dx%d", o->vc_n, o->vc_run);
        FLAG(o->rows_only, " rows_only=%d", o->rows_only); FLAG(o->out_byte_off, " out_off=%zu", o->out_byte_off); FLAG(o->chain_n, " chain=%d", o->chain_n);
        FLAG(o->out_pix_stride, " pix_stride=%d", o->out_pix_stride); FLAG(o->kind == OP_FAIL, " err=%d", o->err);
#undef FLAG
        snprintf(line + k, sizeof line - (size_t)k, "\n");
        EMIT();
    }
    for (uint32_t t = 0; t < m->pub.header.num_tensors; t++) {
        const mtensor_t *mt = &m->mt[t];
        if (!mt->nhwc_c && !mt->partial && !mt->zero_from && !mt->rec_c && !mt->pix_stride) continue;
        snprintf(line, sizeof line, "tensor %u nhwc_c %d pitch %d partial %d zero_from %zu rec_c %d pix_stride %d\n", t, mt->nhwc_c, mt->nhwc_pitch, mt->partial,
                 mt->zero_from, mt->rec_c, mt->pix_stride);
        EMIT();
    }
#undef EMIT
    if (out && cap) out[n < cap ? n : cap - 1] = 0;
    mars_free(&m->pub);
    return n;
}

mars_error_t mars_load_memory(const void *data, size_t size, mars_model_t **model) {
    return mars_hip_load_memory_ex(data, size, 0, model);
}

mars_error_t mars_load_file(const char *path, mars_model_t **model) {
    if (!path || !model) return MARS_ERR_INVALID_FILE;
    FILE *fp = fopen(path, "rb");
    if (!fp) {
        fprintf(stderr, "Mars: Cannot open %s\n", path);
        return MARS_ERR_INVALID_FILE;
    }
    fseek(fp, 0, SEEK_END);
    long size = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    if (size <= 0) { fclose(fp); return MARS_ERR_INVALID_FILE; }
    void *buf = malloc((size_t)size);
    if (!buf) { fclose(fp); return MARS_ERR_ALLOC_FAILED; }
    if (fread(buf, 1, (size_t)size, fp) != (size_t)size) { free(buf); fclose(fp); return MARS_ERR_INVALID_FILE; }
    fclose(fp);
    mars_error_t err = mars_load_memory(buf, (size_t)size, model);
    free(buf); /* the model keeps its own copy (reference :384) */
    return err;
}

/* ---------------------------------------------------------------- accessors */
mars_runtime_tensor_t *mars_get_input(mars_model_t *model, int index) {
    if (!model || index < 0 || (uint32_t)index >= model->header.num_inputs) return NULL;
    uint32_t tid = model->header.input_tensor_ids[index];
    if (tid >= model->header.num_tensors) return NULL;
    return &model->tensors[tid];
}

mars_runtime_tensor_t *mars_get_output(mars_model_t *model, int index) {
    if (!model || index < 0 || (uint32_t)index >= model->header.num_outputs) return NULL;
    uint32_t tid = model->header.output_tensor_ids[index];
    if (tid >= model->header.num_tensors) return NULL;
    return &model->tensors[tid];
}

int mars_get_num_inputs(mars_model_t *model) { return model ? (int)model->header.num_inputs : 0; }
int mars_get_num_outputs(mars_model_t *model) { return model ? (int)model->header.num_outputs : 0; }

void mars_print_summary(mars_model_t *model) {
    if (!model) return;
    mars_model_ext_t *m = (mars_model_ext_t *)model;
    printf("\nMars model summary (MI355X)\n");
    printf("Layers: %u\n", model->header.num_layers);
    printf("Tensors: %u\n", model->header.num_tensors);
    printf("Inputs: %u\n", model->header.num_inputs);
    printf("Outputs: %u\n", model->header.num_outputs);
    printf("Weights: %zu bytes\n", model->weights_size);
    printf("Launches per run: %d, batch %d, activations %zu bytes, parameters %zu bytes\n\n", m->n_ops, m->batch,
           m->act_bytes, m->arena_size);
}

