/*
 * mars_plan.c -- the launch planner of the .mars executor: one planner per layer kind (each cites the reference lines it
 * mirrors: src/mars/mars_runtime.c:511-1224), the host arithmetic they share, the parameter arena, and the fusion passes
 * (SiLU tables, residual Adds, virtual / zero-copy concats, paired launches, pool chains, the float32 policy).
 * Split out of mars_model.c in round 4; loader: mars_model.c, run paths: mars_run.c.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mars_internal.h"
#include "nna.h"

/* ------------------------------------------------------- host arithmetic */
int32_t mars_trunc_x86(float x) {
    /* x86 cvttss2si: out of range or NaN -> INT32_MIN */
    if (x >= -2147483648.0f && x < 2147483648.0f) return (int32_t)x;
    return INT32_MIN;
}
static int sat8(int32_t v) { return v > 127 ? 127 : (v < -128 ? -128 : v); }
static int q_half_up(float v) { return sat8(mars_trunc_x86(v + 0.5f)); }

size_t elem_size(uint32_t dtype) {
    switch (dtype) {
        case MARS_DTYPE_FLOAT32: case MARS_DTYPE_INT32: return 4;
        case MARS_DTYPE_INT16: return 2;
        default: return 1;
    }
}

size_t shape_numel(const mars_tensor_t *d) {
    size_t n = 1;
    for (uint32_t i = 0; i < d->ndims && i < MARS_MAX_DIMS; i++) {
        if (d->shape[i] <= 0) return 0;
        n *= (size_t)d->shape[i];
        if (n > MAX_DIM_PRODUCT) return 0;
    }
    return n;
}

/* Format-aware byte size of a tensor, reference mars_runtime.c:80-124 (`tensor_byte_size`): NDHWC32 rounds the
 * channel count (shape[1]) up to 32, NMHWSOIB2 counts 1024-byte blocks, UINT4 packs two elements per byte; every
 * other tag is numel * element size.  The reference evaluates the products in `int` and converts to size_t; the
 * same wrap-around is kept by computing in 32-bit unsigned arithmetic and sign-extending. */
size_t mars_hip_tensor_byte_size(const mars_tensor_t *t) {
    if (!t) return 0;
    size_t es;
    switch (t->dtype) {
        case MARS_DTYPE_FLOAT32: case MARS_DTYPE_INT32: es = 4; break;
        case MARS_DTYPE_INT16: es = 2; break;
        default: es = 1; break; /* INT8, UINT8, UINT4 (two per byte, handled below), unknown */
    }
    if (t->format == MARS_FORMAT_NDHWC32 && t->ndims >= 4) {
        const uint32_t n = (uint32_t)t->shape[0], h = (uint32_t)t->shape[2], w = (uint32_t)t->shape[3];
        const uint32_t d = (uint32_t)((int32_t)((uint32_t)t->shape[1] + 31u) / 32); /* the reference's int sum, wrapped without UB */
        return (size_t)(int32_t)(n * d * h * w * 32u) * es; /* int product, then * size_t (:101) */
    }
    if (t->format == MARS_FORMAT_NMHWSOIB2 && t->ndims >= 4) {
        const uint32_t no = (uint32_t)((int32_t)((uint32_t)t->shape[0] + 31u) / 32), mi = (uint32_t)((int32_t)((uint32_t)t->shape[1] + 31u) / 32);
        return (size_t)(int32_t)(no * mi * (uint32_t)t->shape[2] * (uint32_t)t->shape[3] * 1024u);
    }
    size_t numel = 1;
    for (uint32_t i = 0; i < t->ndims && i < MARS_MAX_DIMS; i++) numel *= (size_t)(int64_t)t->shape[i];
    if (t->dtype == MARS_DTYPE_UINT4) return (numel + 1) / 2;
    return numel * es;
}

/* What the reference reports as `alloc_size` of every activation tensor (mars_runtime.c:250-334): all of them share
 * working buffers of ONE size, the largest 64-byte-rounded tensor_byte_size of any activation -- or less when three,
 * then two such buffers do not fit behind the weights in its 8 MiB arena.  Callers fill / scan `alloc_size` bytes
 * through vaddr (mars_test.c:73-84, 117-127), so single-frame I/O tensors report the same number here and their
 * staging is at least that large.  Where the reference would refuse to load (weights > 8 MiB, < 64 KiB per buffer)
 * this build still loads and reports the unreduced size. */
size_t reference_buffer_size(const mars_model_ext_t *m) {
    size_t max_sz = 0;
    for (uint32_t i = 0; i < m->pub.header.num_tensors; i++) {
        const mars_tensor_t *d = &m->pub.tensors[i].desc;
        if (d->data_size != 0) continue;
        size_t sz = ALIGN_UP(mars_hip_tensor_byte_size(d), 64);
        if (sz > max_sz) max_sz = sz;
    }
    const size_t ddr = (size_t)8 << 20;
    if (m->pub.weights_size > ddr) return max_sz;
    const size_t remaining = ddr - m->pub.weights_size;
    if (max_sz * 3 <= remaining || max_sz * 2 <= remaining) return max_sz;
    const size_t reduced = (remaining / 2) & ~(size_t)63;
    return reduced >= 65536 ? reduced : max_sz;
}

/* --------------------------------------------------------- parameter arena */
size_t arena_reserve(mars_model_ext_t *m, size_t bytes) {
    /* any failure here fails the whole plan (build_plan returns plan_err): no op keeps an unset offset */
    if (bytes > ARENA_MAX || m->arena_size > ARENA_MAX) { m->plan_err = MARS_ERR_ALLOC_FAILED; return NO_OFF; }
    size_t off = ALIGN_UP(m->arena_size, 256);
    size_t end = off + ALIGN_UP(bytes ? bytes : 1, 256);
    if (end > ARENA_MAX) { m->plan_err = MARS_ERR_ALLOC_FAILED; return NO_OFF; }
    if (end > m->arena_cap) {
        size_t cap = m->arena_cap ? m->arena_cap : (1u << 20);
        while (cap < end) cap *= 2; /* end <= 2^40: cannot overflow */
        uint8_t *p = (uint8_t *)realloc(m->arena_host, cap);
        if (!p) { m->plan_err = MARS_ERR_ALLOC_FAILED; return NO_OFF; }
        memset(p + m->arena_cap, 0, cap - m->arena_cap);
        m->arena_host = p;
        m->arena_cap = cap;
    }
    m->arena_size = end;
    return off;
}

/* bytes [off, off+n) of the weight blob; beyond its end the blob reads as zeros */
void blob_read(const mars_model_ext_t *m, size_t off, size_t n, void *dst) {
    const uint8_t *blob = (const uint8_t *)m->pub.weights;
    size_t have = m->pub.weights_size;
    memset(dst, 0, n);
    if (!blob || off >= have) return;
    size_t c = have - off < n ? have - off : n;
    memcpy(dst, blob + off, c);
}

/* weights -> [oc_pad][k64] rows, each kernel row padded to row_pad bytes.
 * nchw: source is OIHW and the (transposed) input has c_pad channels. */
void mars_pack_conv_i8(const int8_t *w, size_t avail, int nchw, int out_c, int in_c, int kh, int kw,
                       int c_pad, int row_pad, int oc_pad, int8_t *dst) {
    const int k64 = (int)ALIGN_UP((size_t)kh * row_pad, 64);
    memset(dst, 0, (size_t)oc_pad * k64);
    for (int oc = 0; oc < out_c; oc++)
        for (int ky = 0; ky < kh; ky++)
            for (int kx = 0; kx < kw; kx++)
                for (int ic = 0; ic < in_c; ic++) {
                    size_t src = nchw ? (((size_t)oc * in_c + ic) * kh + ky) * kw + kx
                                      : (((size_t)oc * kh + ky) * kw + kx) * in_c + ic;
                    int8_t v = src < avail ? w[src] : 0;
                    dst[(size_t)mhip_conv_i8_oc_row(oc, oc_pad) * k64 + (size_t)ky * row_pad + (size_t)kx * c_pad + ic] = v;
                }
}

/* ------------------------------------------------------------------- ops */
static mars_op_t *new_op(mars_model_ext_t *m, int kind, int layer) {
    if (m->n_ops == m->cap_ops) {
        int cap = m->cap_ops ? m->cap_ops * 2 : 64;
        mars_op_t *p = (mars_op_t *)realloc(m->ops, (size_t)cap * sizeof(mars_op_t));
        if (!p) { m->plan_err = MARS_ERR_ALLOC_FAILED; return NULL; }
        m->ops = p;
        m->cap_ops = cap;
    }
    mars_op_t *op = &m->ops[m->n_ops++];
    memset(op, 0, sizeof(*op));
    op->kind = kind;
    op->layer = layer;
    op->t_in[0] = op->t_in[1] = op->t_in[2] = op->t_in[3] = op->t_out = -1;
    op->w_off = op->b_off = op->lut_off = op->lut2_off = op->s_off = op->w2_off = op->w3_off = NO_OFF;
    op->w_blob_off[0] = op->w_blob_off[1] = NO_OFF;
    op->prof_kind = 4;
    return op;
}

static void fail_op(mars_model_ext_t *m, int layer, int err) {
    mars_op_t *op = new_op(m, OP_FAIL, layer);
    if (op) op->err = err;
}

/* the executor searches tensors by desc.id, first match (mars_runtime.c:516-558, 713-721) */
static int find_tensor(const mars_model_ext_t *m, uint32_t id) {
    if (id == NO_TENSOR) return -1;
    for (uint32_t i = 0; i < m->pub.header.num_tensors; i++)
        if (m->pub.tensors[i].desc.id == id) return (int)i;
    return -1;
}

static void touch(mars_model_ext_t *m, int ti, size_t extent) {
    if (ti < 0) return;
    mtensor_t *t = &m->mt[ti];
    t->needed = 1;
    if (t->is_weight) {
        size_t end = (size_t)m->pub.tensors[ti].desc.data_offset + extent;
        if (end > m->blob_mirror_bytes) m->blob_mirror_bytes = end;
    } else if (extent > t->extent) {
        t->extent = extent;
    }
}

static size_t lut_i8(mars_model_ext_t *m, const int8_t table[256]) {
    size_t off = arena_reserve(m, 256);
    if (off != NO_OFF) memcpy(m->arena_host + off, table, 256);
    return off;
}

/* int8 sigmoid of one value, reference mars_runtime.c:758-768 */
static int sigmoid_q(int q, float in_scale, float out_scale) {
    float os = out_scale > 0 ? out_scale : 1.0f;
    float x = (float)q * in_scale;
    float y = 1.0f / (1.0f + expf(-x));
    return q_half_up(y / os);
}
/* int8 mul/add of one pair, reference mars_runtime.c:822-835 / :889-902 */
static int binary_q(int is_mul, int a, int b, float sa, float sb, float so) {
    float inv = 1.0f / (so > 0 ? so : 1.0f);
    float va = (float)a * sa, vb = (float)b * sb;
    float y = is_mul ? va * vb : va + vb;
    return q_half_up(y * inv);
}

/* ---- CONV2D: reference mars_runtime.c:511-710 */
static void plan_conv(mars_model_ext_t *m, int li) {
    const mars_layer_t *L = &m->pub.layers[li].desc;
    const mars_conv_params_t *cp = &L->params.conv;
    int ti = find_tensor(m, L->input_tensor_ids[0]);
    int to = find_tensor(m, L->output_tensor_ids[0]);
    int tw = find_tensor(m, cp->weight_tensor_id);
    int tb = find_tensor(m, cp->bias_tensor_id);
    if (ti < 0 || to < 0 || tw < 0) { fail_op(m, li, MARS_ERR_INVALID_TENSOR); return; }
    const mars_tensor_t *in = &m->pub.tensors[ti].desc, *out = &m->pub.tensors[to].desc;
    const mars_tensor_t *w = &m->pub.tensors[tw].desc;
    /* weights / bias must come from the blob, the result must be an activation, and a
     * parallel kernel cannot run a layer in place */
    if (!m->mt[tw].is_weight || (tb >= 0 && !m->mt[tb].is_weight) || m->mt[to].is_weight || ti == to) {
        fail_op(m, li, MARS_ERR_LAYER_FAILED);
        return;
    }
    const int in_nhwc = in->format == MARS_FORMAT_NHWC, out_nhwc = out->format == MARS_FORMAT_NHWC;
    int in_h, in_w, in_c, out_h, out_w, out_c;
    if (in_nhwc) { in_h = in->shape[1]; in_w = in->shape[2]; in_c = in->shape[3]; }
    else         { in_c = in->shape[1]; in_h = in->shape[2]; in_w = in->shape[3]; }
    if (out_nhwc) { out_h = out->shape[1]; out_w = out->shape[2]; out_c = out->shape[3]; }
    else          { out_c = out->shape[1]; out_h = out->shape[2]; out_w = out->shape[3]; }
    const int kh = (int)cp->kernel_h, kw = (int)cp->kernel_w, sh = (int)cp->stride_h, sw = (int)cp->stride_w;
    int pt = 0, pl = 0;
    if (cp->padding == MARS_PAD_SAME) { /* EXPLICIT / VALID run unpadded (:592-598) */
        int32_t ph = (int32_t)((uint32_t)(out_h - 1) * cp->stride_h + cp->kernel_h - (uint32_t)in_h);
        int32_t pw = (int32_t)((uint32_t)(out_w - 1) * cp->stride_w + cp->kernel_w - (uint32_t)in_w);
        pt = ph / 2;
        pl = pw / 2;
    }
    if (out_h <= 0 || out_w <= 0 || out_c <= 0) return; /* empty loops in the reference: nothing is written */
    /* channel counts are bounded so that every packed-weight product below (kw * c_pad, kh * row_pad, oc_pad * k64)
     * stays far inside int / size_t: a crafted in_c once wrapped kw * in_c and let the packer write past its slot */
    if (in_h <= 0 || in_w <= 0 || in_c <= 0 || kh <= 0 || kw <= 0 || sh < 0 || sw < 0 || kh > 64 || kw > 64 ||
        in_c > MAX_CHANNELS || out_c > MAX_CHANNELS ||
        (size_t)in_h * in_w * in_c > MAX_DIM_PRODUCT || (size_t)out_h * out_w * out_c > MAX_DIM_PRODUCT ||
        (size_t)out_c * in_c * kh * kw > ((size_t)1 << 31)) {
        fail_op(m, li, MARS_ERR_LAYER_FAILED); /* degenerate geometry this build does not launch */
        return;
    }
    const int is_f32 = in->dtype == MARS_DTYPE_FLOAT32;
    mars_op_t *op = new_op(m, is_f32 ? OP_CONV_F32 : OP_CONV_I8, li);
    if (!op) return;
    op->t_in[0] = ti; op->n_in = 1; op->t_out = to;
    op->in_h = in_h; op->in_w = in_w; op->in_c = in_c;
    op->out_h = out_h; op->out_w = out_w; op->out_c = out_c;
    op->kh = kh; op->kw = kw; op->sh = sh; op->sw = sw; op->pt = pt; op->pl = pl;
    op->is_f32 = is_f32;
    op->macs = (double)out_h * out_w * out_c * in_c * kh * kw;
    const size_t es = is_f32 ? 4 : 1;
    touch(m, ti, (size_t)in_h * in_w * in_c * es);
    touch(m, to, (size_t)out_h * out_w * out_c * es);
    m->mt[tw].needed = 1;
    op->bytes = (double)((size_t)in_h * in_w * in_c + (size_t)out_h * out_w * out_c) * es;
    const size_t wcount = (size_t)out_c * in_c * kh * kw;

    if (is_f32) {
        op->prof_kind = 1;
        op->w_off = arena_reserve(m, wcount * 4);
        if (op->w_off == NO_OFF) return;
        if (!m->deferred) blob_read(m, (size_t)w->data_offset, wcount * 4, m->arena_host + op->w_off);
        /* The bf16 images of the weights the split-operand matrix-core kernels read, for the f32_mfma mode in force AT LOAD (set it
         * before loading; a model switched to mode 3 / 4 later runs its convolutions on the f32 matrix cores instead): mode 3 =
         * conv_f32_patch's image where the shape is one it takes (k x k layers; unit table + two planes in its K order), else two
         * planes for conv_f32_split; mode 4 = three planes for conv_f32_split; modes 0 - 2 read neither (no dead planes in
         * the arena, its upload and its multi-GPU broadcast). */
        {
            const int mode = mhip_conv_f32_mode(-1);
            size_t n3 = 0;
            int stem = 0;
            if (mode == 3 && sh == sw && pt == pl) {
                n3 = mhip_conv_f32_stem_pack(out_c, in_c, kh, kw, sw, pl, in_h, in_w, out_h, out_w, NULL, NULL);
                stem = n3 != 0;
                if (!n3) n3 = mhip_conv_f32_patch_pack(out_c, in_c, kh, kw, sw, pl, in_h, in_w, out_h, out_w, NULL, NULL);
            }
            if (n3) {
                op->w3_stem = stem;
                op->w3_off = arena_reserve(m, n3);
                if (op->w3_off == NO_OFF) return;
                if (!m->deferred) {
                    if (stem)
                        mhip_conv_f32_stem_pack(out_c, in_c, kh, kw, sw, pl, in_h, in_w, out_h, out_w, (const float *)(m->arena_host + op->w_off),
                                                m->arena_host + op->w3_off);
                    else
                        mhip_conv_f32_patch_pack(out_c, in_c, kh, kw, sw, pl, in_h, in_w, out_h, out_w, (const float *)(m->arena_host + op->w_off),
                                                 m->arena_host + op->w3_off);
                }
            } else if (mode >= 3) {
                const int planes = mode == 3 ? 2 : 3;
                const size_t n2 = mhip_conv_f32_split_pack(out_c, in_c, kh, kw, sw, planes, NULL, NULL);
                if (n2) {
                    op->w2_off = arena_reserve(m, n2);
                    if (op->w2_off == NO_OFF) return;
                    op->w2_planes = planes; /* what a later mode change may read (conv_f32_params) */
                    if (!m->deferred)
                        mhip_conv_f32_split_pack(out_c, in_c, kh, kw, sw, planes, (const float *)(m->arena_host + op->w_off), m->arena_host + op->w2_off);
                }
            }
        }
        if (tb >= 0) {
            op->b_off = arena_reserve(m, (size_t)out_c * 4);
            if (op->b_off != NO_OFF && !m->deferred)
                blob_read(m, (size_t)m->pub.tensors[tb].desc.data_offset, (size_t)out_c * 4, m->arena_host + op->b_off);
        }
        if (cp->activation == MARS_ACT_RELU) { /* byte-wise clamp over H*W*C BYTES of the f32 result (:700-707) */
            mars_op_t *r = new_op(m, OP_RELU_BYTES, li);
            if (r) { r->t_out = to; r->n = (size_t)out_h * out_w * out_c; r->prof_kind = 2; r->bytes = 2.0 * r->n; }
        }
        return;
    }

    op->prof_kind = 0;
    op->nchw = !in_nhwc; /* the kernel is chosen by the INPUT tag (:640-662) */
    op->out_nchw = op->nchw; /* ... and conv2d_int8_mxu writes [O][H][W] whatever the output's tag says */
    op->c_pad = op->nchw ? (int)ALIGN_UP((size_t)in_c, 16) : in_c;
    /* NCHW stem (round 6: the shipped files' first layer is 3 planes of 640 x 640): relaid to 4 bytes per pixel, the small-channel kernels'
     * own input layout (K = 6 rows x 32 bytes), instead of 16 (K = 6 x 96: 2.4 of yolov5n_int8.mars' 12.5 ms per batch went there) */
    if (op->nchw && in_c <= 4 && mhip_conv_i8_small_c(4, kw, out_c) && ((size_t)in_h * in_w) % 4 == 0) op->c_pad = 4;
    int c_eff = op->c_pad; /* bytes per pixel in the packed K layout (4 in small-channel mode) */
    mhip_conv_i8_pack_geom(op->c_pad, kw, out_c, &op->row_pad, &op->oc_pad, &c_eff);
    const size_t k64 = ALIGN_UP((size_t)kh * op->row_pad, 64);
    op->w_off = arena_reserve(m, (size_t)op->oc_pad * k64);
    if (op->w_off == NO_OFF) return;
    if (!m->deferred) {
        int8_t *tmp = (int8_t *)malloc(wcount ? wcount : 1);
        if (!tmp) { m->plan_err = MARS_ERR_ALLOC_FAILED; return; }
        blob_read(m, (size_t)w->data_offset, wcount, tmp);
        mars_pack_conv_i8(tmp, wcount, op->nchw, out_c, in_c, kh, kw, c_eff, op->row_pad, op->oc_pad,
                          (int8_t *)m->arena_host + op->w_off);
        free(tmp);
    }
    if (!op->nchw) { /* RGB stem: its matrix-core operands pre-laid for conv_i8_rgb (12 KB for 32 channels, 6 x 6) */
        const size_t n2 = mhip_conv_i8_rgb_pack(in_c, kh, kw, sh, sw, pl, op->oc_pad, (int)k64, NULL, NULL);
        if (n2) {
            op->w2_off = arena_reserve(m, n2);
            if (op->w2_off == NO_OFF) return;
            if (!m->deferred)
                mhip_conv_i8_rgb_pack(in_c, kh, kw, sh, sw, pl, op->oc_pad, (int)k64, (const int8_t *)m->arena_host + op->w_off,
                                      (int8_t *)m->arena_host + op->w2_off);
        }
    }
    if (!op->nchw && op->w2_off == NO_OFF) { /* deep 3x3 stride-1 layers: the weight image conv_i8_rows streams (same bytes, K-step blocks) */
        const size_t n3 = mhip_conv_i8_rows_pack(in_c, kh, kw, sh, sw, op->oc_pad, (int)k64, out_w, NULL, NULL);
        if (n3) {
            op->w2_off = arena_reserve(m, n3);
            if (op->w2_off == NO_OFF) return;
            op->w2_rows = 1;
            if (!m->deferred)
                mhip_conv_i8_rows_pack(in_c, kh, kw, sh, sw, op->oc_pad, (int)k64, out_w, (const int8_t *)m->arena_host + op->w_off,
                                       (int8_t *)m->arena_host + op->w2_off);
        }
    }
    if (tb >= 0) { /* raw bytes reinterpreted as int32, whatever the tensor says it is (:645,656) */
        op->b_off = arena_reserve(m, (size_t)op->oc_pad * 4);
        if (op->b_off != NO_OFF && !m->deferred) {
            int32_t *raw = (int32_t *)calloc((size_t)out_c, 4), *dstb = (int32_t *)(m->arena_host + op->b_off);
            if (!raw) { m->plan_err = MARS_ERR_ALLOC_FAILED; return; }
            blob_read(m, (size_t)m->pub.tensors[tb].desc.data_offset, (size_t)out_c * 4, raw);
            for (int oc = 0; oc < out_c; oc++) dstb[mhip_conv_i8_oc_row(oc, op->oc_pad)] = raw[oc]; /* same row order as the weights */
            free(raw);
        }
    }
    op->cs = (in->scale * w->scale) / out->scale; /* float32, this order (mxu_conv.c:639,722) */
    op->safe = mhip_conv_i8_is_safe(op->cs);
    op->relu = cp->activation == MARS_ACT_RELU;
    if (op->nchw) {
        size_t need = (size_t)in_h * in_w * op->c_pad;
        if (need > m->scratch_per_frame) m->scratch_per_frame = need;
        /* (op->bytes stays input + output: the relayout's traffic is this implementation's cost, not algorithmic work) */
    }
}

static size_t numel_of(const mars_model_ext_t *m, int ti) { return shape_numel(&m->pub.tensors[ti].desc); }

/* ---- SIGMOID / RELU family: unary maps (mars_runtime.c:724-771, 1047-1089) */
static void plan_unary(mars_model_ext_t *m, int li) {
    const mars_layer_t *L = &m->pub.layers[li].desc;
    int ti = find_tensor(m, L->input_tensor_ids[0]), to = find_tensor(m, L->output_tensor_ids[0]);
    if (ti < 0 || to < 0) { fail_op(m, li, MARS_ERR_INVALID_TENSOR); return; }
    if (m->mt[to].is_weight) { fail_op(m, li, MARS_ERR_LAYER_FAILED); return; }
    const mars_tensor_t *in = &m->pub.tensors[ti].desc, *out = &m->pub.tensors[to].desc;
    const size_t n = numel_of(m, ti);
    if (n == 0) return;
    const int is_sig = L->type == MARS_LAYER_SIGMOID;
    const int leaky = L->type == MARS_LAYER_LEAKY_RELU;
    if (in->dtype == MARS_DTYPE_FLOAT32) {
        mars_op_t *op = new_op(m, is_sig ? OP_SIGMOID_F32 : OP_RELU_F32, li);
        if (!op) return;
        op->t_in[0] = ti; op->n_in = 1; op->t_out = to; op->n = n;
        op->f0 = leaky ? 0.01f : 0.0f; /* slope is a constant in the reference (:1064) */
        op->prof_kind = 2; op->bytes = 8.0 * n;
        touch(m, ti, n * 4); touch(m, to, n * 4);
        return;
    }
    int8_t tab[256];
    for (int q = -128; q < 128; q++) {
        int r;
        if (is_sig) r = sigmoid_q(q, in->scale, out->scale);
        else if (q > 0) r = q;
        else if (leaky) { int32_t v = mars_trunc_x86((float)q * 0.01f); r = (int8_t)(v < -128 ? -128 : v); }
        else r = 0; /* RELU and RELU6 alike: no upper clamp (:1179-1182) */
        tab[q + 128] = (int8_t)r;
    }
    mars_op_t *op = new_op(m, OP_LUT_I8, li);
    if (!op) return;
    op->t_in[0] = ti; op->n_in = 1; op->t_out = to; op->n = n;
    op->lut_off = lut_i8(m, tab);
    op->prof_kind = 2; op->bytes = 2.0 * n;
    touch(m, ti, n); touch(m, to, n);
}

/* ---- MUL / ADD (mars_runtime.c:774-905): extent from operand A only */
static void plan_binary(mars_model_ext_t *m, int li) {
    const mars_layer_t *L = &m->pub.layers[li].desc;
    int ta = find_tensor(m, L->input_tensor_ids[0]), tb = find_tensor(m, L->input_tensor_ids[1]);
    int to = find_tensor(m, L->output_tensor_ids[0]);
    if (ta < 0 || tb < 0 || to < 0) { fail_op(m, li, MARS_ERR_INVALID_TENSOR); return; }
    if (m->mt[to].is_weight) { fail_op(m, li, MARS_ERR_LAYER_FAILED); return; }
    const mars_tensor_t *a = &m->pub.tensors[ta].desc, *b = &m->pub.tensors[tb].desc, *o = &m->pub.tensors[to].desc;
    const size_t n = numel_of(m, ta);
    if (n == 0) return;
    const int f32 = a->dtype == MARS_DTYPE_FLOAT32;
    mars_op_t *op = new_op(m, f32 ? OP_BINARY_F32 : OP_BINARY_I8, li);
    if (!op) return;
    op->t_in[0] = ta; op->t_in[1] = tb; op->n_in = 2; op->t_out = to; op->n = n;
    op->is_mul = L->type == MARS_LAYER_MUL;
    op->f0 = a->scale; op->f1 = b->scale;
    op->f2 = 1.0f / (o->scale > 0 ? o->scale : 1.0f);
    op->prof_kind = 2; op->bytes = 3.0 * n * (f32 ? 4 : 1);
    const size_t es = f32 ? 4 : 1;
    touch(m, ta, n * es); touch(m, tb, n * es); touch(m, to, n * es);
}

/* ---- MAXPOOL (mars_runtime.c:908-960) */
static void plan_maxpool(mars_model_ext_t *m, int li) {
    const mars_layer_t *L = &m->pub.layers[li].desc;
    const mars_pool_params_t *pp = &L->params.pool;
    int ti = find_tensor(m, L->input_tensor_ids[0]), to = find_tensor(m, L->output_tensor_ids[0]);
    if (ti < 0 || to < 0) { fail_op(m, li, MARS_ERR_INVALID_TENSOR); return; }
    if (m->mt[to].is_weight || ti == to) { fail_op(m, li, MARS_ERR_LAYER_FAILED); return; }
    const mars_tensor_t *in = &m->pub.tensors[ti].desc, *out = &m->pub.tensors[to].desc;
    int in_h = in->shape[1], in_w = in->shape[2], ch = in->shape[3], out_h = out->shape[1], out_w = out->shape[2];
    if (out_h <= 0 || out_w <= 0 || ch <= 0) return;
    if (in_h < 0 || in_w < 0 || (int)pp->kernel_h < 0 || (int)pp->kernel_w < 0 || (int)pp->stride_h < 0 ||
        (int)pp->stride_w < 0 || pp->kernel_h > 4096 || pp->kernel_w > 4096) {
        fail_op(m, li, MARS_ERR_LAYER_FAILED);
        return;
    }
    mars_op_t *op = new_op(m, OP_MAXPOOL, li);
    if (!op) return;
    op->t_in[0] = ti; op->n_in = 1; op->t_out = to;
    op->in_h = in_h; op->in_w = in_w; op->in_c = ch; op->out_h = out_h; op->out_w = out_w;
    op->kh = (int)pp->kernel_h; op->kw = (int)pp->kernel_w; op->sh = (int)pp->stride_h; op->sw = (int)pp->stride_w;
    op->prof_kind = 3;
    op->bytes = (double)in_h * in_w * ch + (double)out_h * out_w * ch;
    /* reads stay inside in_h*in_w*ch by the window clip; writes cover out_h*out_w*ch */
    if (op->kh > 0 && op->kw > 0) touch(m, ti, (size_t)in_h * in_w * ch);
    else m->mt[ti].needed = 1;
    touch(m, to, (size_t)out_h * out_w * ch);
}

/* ---- CONCAT (mars_runtime.c:963-1000): one copy kernel per input, in order */
static void plan_concat(mars_model_ext_t *m, int li) {
    const mars_layer_t *L = &m->pub.layers[li].desc;
    int to = find_tensor(m, L->output_tensor_ids[0]);
    if (to < 0) { fail_op(m, li, MARS_ERR_INVALID_TENSOR); return; }
    if (m->mt[to].is_weight) { fail_op(m, li, MARS_ERR_LAYER_FAILED); return; }
    const mars_tensor_t *out = &m->pub.tensors[to].desc;
    const int out_h = out->shape[1], out_w = out->shape[2], out_c = out->shape[3];
    if (L->num_inputs > 4) { fail_op(m, li, MARS_ERR_INVALID_LAYER); return; }
    int off = 0;
    for (uint32_t k = 0; k < L->num_inputs; k++) {
        int ti = find_tensor(m, L->input_tensor_ids[k]);
        if (ti < 0) continue; /* skipped without advancing the channel offset (:980) */
        const int in_c = m->pub.tensors[ti].desc.shape[3];
        if (out_h > 0 && out_w > 0 && in_c > 0) {
            /* pixel slots of one input may not overlap each other in a parallel copy */
            if (in_c > out_c || out_c <= 0 || ti == to) { fail_op(m, li, MARS_ERR_LAYER_FAILED); return; }
            mars_op_t *op = new_op(m, OP_CONCAT_SLICE, li);
            if (!op) return;
            op->t_in[0] = ti; op->n_in = 1; op->t_out = to;
            op->out_h = out_h; op->out_w = out_w; op->in_c = in_c; op->out_c = out_c; op->ch_off = off;
            op->prof_kind = 3;
            const size_t npix = (size_t)out_h * out_w;
            op->bytes = 2.0 * npix * in_c;
            touch(m, ti, npix * in_c);
            touch(m, to, (npix - 1) * out_c + off + in_c);
        }
        off += in_c;
    }
}

/* ---- UPSAMPLE (mars_runtime.c:1003-1044) */
static void plan_upsample(mars_model_ext_t *m, int li) {
    const mars_layer_t *L = &m->pub.layers[li].desc;
    const mars_upsample_params_t *up = &L->params.upsample;
    int ti = find_tensor(m, L->input_tensor_ids[0]), to = find_tensor(m, L->output_tensor_ids[0]);
    if (ti < 0 || to < 0) { fail_op(m, li, MARS_ERR_INVALID_TENSOR); return; }
    if (m->mt[to].is_weight || ti == to) { fail_op(m, li, MARS_ERR_LAYER_FAILED); return; }
    const mars_tensor_t *in = &m->pub.tensors[ti].desc, *out = &m->pub.tensors[to].desc;
    int in_h = in->shape[1], in_w = in->shape[2], ch = in->shape[3], out_h = out->shape[1], out_w = out->shape[2];
    if ((up->scale_h == 0 && in_h == 0) || (up->scale_w == 0 && in_w == 0)) { fail_op(m, li, MARS_ERR_LAYER_FAILED); return; }
    int sh = up->scale_h > 0 ? (int)up->scale_h : out_h / in_h;
    int sw = up->scale_w > 0 ? (int)up->scale_w : out_w / in_w;
    if (out_h <= 0 || out_w <= 0 || ch <= 0) return;
    if (sh <= 0 || sw <= 0 || in_h <= 0 || in_w <= 0) { fail_op(m, li, MARS_ERR_LAYER_FAILED); return; }
    mars_op_t *op = new_op(m, OP_UPSAMPLE, li);
    if (!op) return;
    op->t_in[0] = ti; op->n_in = 1; op->t_out = to;
    op->in_h = in_h; op->in_w = in_w; op->in_c = ch; op->out_h = out_h; op->out_w = out_w;
    op->scale_h = sh; op->scale_w = sw;
    op->prof_kind = 3;
    op->bytes = (double)in_h * in_w * ch + (double)out_h * out_w * ch;
    touch(m, ti, (size_t)in_h * in_w * ch);
    touch(m, to, (size_t)out_h * out_w * ch);
}

/* ---- BATCHNORM (mars_runtime.c:1092-1158) */
static void plan_batchnorm(mars_model_ext_t *m, int li) {
    const mars_layer_t *L = &m->pub.layers[li].desc;
    int ti = find_tensor(m, L->input_tensor_ids[0]), to = find_tensor(m, L->output_tensor_ids[0]);
    int ts = find_tensor(m, L->input_tensor_ids[1]), tb = find_tensor(m, L->input_tensor_ids[2]);
    if (ti < 0 || to < 0) { fail_op(m, li, MARS_ERR_INVALID_TENSOR); return; }
    if (m->mt[to].is_weight || (ts >= 0 && !m->mt[ts].is_weight) || (tb >= 0 && !m->mt[tb].is_weight)) {
        fail_op(m, li, MARS_ERR_LAYER_FAILED);
        return;
    }
    const mars_tensor_t *in = &m->pub.tensors[ti].desc, *out = &m->pub.tensors[to].desc;
    int n = in->shape[0] > 0 ? in->shape[0] : 1, c = in->shape[1] > 0 ? in->shape[1] : 1;
    int h = in->shape[2] > 0 ? in->shape[2] : 1, w = in->shape[3] > 0 ? in->shape[3] : 1;
    if ((size_t)n * c * h * w > MAX_DIM_PRODUCT || c > MAX_CHANNELS) { fail_op(m, li, MARS_ERR_LAYER_FAILED); return; }
    const int f32 = in->dtype == MARS_DTYPE_FLOAT32;
    mars_op_t *op = new_op(m, OP_BN, li);
    if (!op) return;
    op->t_in[0] = ti; op->n_in = 1; op->t_out = to;
    op->bn_n = n; op->in_c = c; op->in_h = h; op->in_w = w; op->is_f32 = f32;
    op->f0 = in->scale > 0 ? in->scale : 1.0f;
    op->f1 = out->scale > 0 ? out->scale : 1.0f;
    if (ts >= 0) {
        op->s_off = arena_reserve(m, (size_t)c * 4);
        if (op->s_off != NO_OFF && !m->deferred)
            blob_read(m, (size_t)m->pub.tensors[ts].desc.data_offset, (size_t)c * 4, m->arena_host + op->s_off);
    }
    if (tb >= 0) {
        op->b_off = arena_reserve(m, (size_t)c * 4);
        if (op->b_off != NO_OFF && !m->deferred)
            blob_read(m, (size_t)m->pub.tensors[tb].desc.data_offset, (size_t)c * 4, m->arena_host + op->b_off);
    }
    const size_t total = (size_t)n * c * h * w * (f32 ? 4 : 1);
    op->prof_kind = 2; op->bytes = 2.0 * total;
    touch(m, ti, total); touch(m, to, total);
}

/* dispatcher, reference mars_runtime.c:1161-1224 */
void plan_layer(mars_model_ext_t *m, int li) {
    switch (m->pub.layers[li].desc.type) {
        case MARS_LAYER_CONV2D: plan_conv(m, li); break;
        case MARS_LAYER_SIGMOID:
        case MARS_LAYER_RELU:
        case MARS_LAYER_RELU6:
        case MARS_LAYER_LEAKY_RELU: plan_unary(m, li); break;
        case MARS_LAYER_MUL:
        case MARS_LAYER_ADD: plan_binary(m, li); break;
        case MARS_LAYER_MAXPOOL: plan_maxpool(m, li); break;
        case MARS_LAYER_CONCAT: plan_concat(m, li); break;
        case MARS_LAYER_UPSAMPLE: plan_upsample(m, li); break;
        case MARS_LAYER_BATCHNORM: plan_batchnorm(m, li); break;
        case MARS_LAYER_DEPTHWISE_CONV2D: /* accepted, not executed (:1168-1213) */
        case MARS_LAYER_AVGPOOL:
        case MARS_LAYER_SILU:
        case MARS_LAYER_RESHAPE:
        case MARS_LAYER_TRANSPOSE:
        case MARS_LAYER_SOFTMAX: break;
        default: fail_op(m, li, MARS_ERR_INVALID_LAYER); break; /* GLOBAL_AVGPOOL, FC, unknown (:1218-1220) */
    }
}

/* ------------------------------------------------------------------ fusion
 * conv -> sigmoid -> mul (SiLU as exported to ONNX) collapses into the conv
 * epilogue: every value of the chain is a function of the conv's int8 result
 * q1 alone, so lut[q1] = mul(q1, sigmoid(q1)) computed on the host with the
 * reference's own float steps is bit-identical.
 * Only done when q1 and sigmoid(q1) have no other reader and are not outputs. */
void fuse_silu(mars_model_ext_t *m) {
    const int nt = (int)m->pub.header.num_tensors;
    int *readers = (int *)calloc((size_t)nt + 1, sizeof(int));
    int *writers = (int *)calloc((size_t)nt + 1, sizeof(int));
    if (!readers || !writers) { free(readers); free(writers); return; }
    for (int i = 0; i < m->n_ops; i++) {
        for (int k = 0; k < m->ops[i].n_in; k++)
            if (m->ops[i].t_in[k] >= 0) readers[m->ops[i].t_in[k]]++;
        if (m->ops[i].t_out >= 0) writers[m->ops[i].t_out]++;
    }
    for (int i = 0; i + 2 < m->n_ops; i++) {
        mars_op_t *c = &m->ops[i], *s = &m->ops[i + 1], *mu = &m->ops[i + 2];
        if (c->kind != OP_CONV_I8 || s->kind != OP_LUT_I8 || mu->kind != OP_BINARY_I8 || !mu->is_mul) continue;
        if (m->pub.layers[s->layer].desc.type != MARS_LAYER_SIGMOID) continue;
        const int q1 = c->t_out, q2 = s->t_out, q3 = mu->t_out;
        if (s->t_in[0] != q1) continue;
        const int fwd = mu->t_in[0] == q1 && mu->t_in[1] == q2, rev = mu->t_in[0] == q2 && mu->t_in[1] == q1;
        if (!fwd && !rev) continue;
        if (q1 == q2 || q2 == q3 || q1 == q3) continue;
        { /* redirecting the result onto one of the convolution's own inputs would make a parallel launch run in place */
            int inplace = 0;
            for (int k = 0; k < c->n_in; k++)
                if (c->t_in[k] == q3) inplace = 1;
            if (inplace) continue;
        }
        if (readers[q1] != 2 || readers[q2] != 1 || writers[q1] != 1 || writers[q2] != 1 || writers[q3] != 1) continue;
        if (m->mt[q1].io_out || m->mt[q2].io_out || m->mt[q1].io_in || m->mt[q2].io_in) continue;
        const size_t n1 = (size_t)c->out_h * c->out_w * c->out_c;
        if (s->n != n1 || mu->n != n1) continue; /* the chain must cover exactly the conv's result */
        const mars_tensor_t *d1 = &m->pub.tensors[q1].desc, *d2 = &m->pub.tensors[q2].desc, *d3 = &m->pub.tensors[q3].desc;
        int8_t tab[256];
        for (int q = -128; q < 128; q++) {
            int sg = sigmoid_q(q, d1->scale, d2->scale);
            tab[q + 128] = (int8_t)(fwd ? binary_q(1, q, sg, d1->scale, d2->scale, d3->scale)
                                        : binary_q(1, sg, q, d2->scale, d1->scale, d3->scale));
        }
        c->lut_off = lut_i8(m, tab);
        if (mhip_conv_i8_lut2_ok(c->cs)) {
            /* half-step form: index k = trunc(2 * acc * cs) in [-256, 255] determines round-half-away(acc * cs) =
             * (k >= 0 ? (k+1)>>1 : -((1-k)>>1)) exactly (conv_i8.hip, requant_pack FAST); the ReLU clamp folds in */
            int8_t tab2[512];
            for (int k = -256; k < 256; k++) {
                int r = k >= 0 ? (k + 1) >> 1 : -((1 - k) >> 1);
                const int lo = c->relu ? 0 : -128;
                r = r < lo ? lo : (r > 127 ? 127 : r);
                tab2[k + 256] = tab[r + 128];
            }
            c->lut2_off = arena_reserve(m, 512);
            if (c->lut2_off != NO_OFF) memcpy(m->arena_host + c->lut2_off, tab2, 512);
        }
        c->t_out = q3;
        c->bytes += 0; /* same bytes written, to q3 instead of q1 */
        touch(m, q3, n1);
        m->mt[q1].needed = 0;
        m->mt[q2].needed = 0;
        /* drop the two element-wise ops */
        memmove(&m->ops[i + 1], &m->ops[i + 3], (size_t)(m->n_ops - i - 3) * sizeof(mars_op_t));
        m->n_ops -= 2;
    }
    free(readers);
    free(writers);
}

/* A single int8 activation layer behind a convolution (RELU / RELU6 / LEAKY_RELU / SIGMOID as their own layers: the shipped tiny_160_int8.mars,
 * any exporter that does not fold them): the layer is a 256-entry map of the convolution's int8 result (plan_unary builds it), which is what the
 * convolution's fused table applies in its epilogue -- the same bytes, one launch and two passes over the tensor less.  Runs after fuse_silu (a
 * SIGMOID that feeds a MUL belongs to that chain).  Conditions as there: the convolution's result has no other reader and is no graph tensor. */
void fuse_lut(mars_model_ext_t *m) {
    if (getenv("MARS_HIP_NO_FUSE_LUT")) return;
    const int nt = (int)m->pub.header.num_tensors;
    int *readers = (int *)calloc((size_t)nt + 1, sizeof(int));
    int *writers = (int *)calloc((size_t)nt + 1, sizeof(int));
    if (!readers || !writers) { free(readers); free(writers); return; }
    for (int i = 0; i < m->n_ops; i++) {
        for (int k = 0; k < m->ops[i].n_in; k++)
            if (m->ops[i].t_in[k] >= 0) readers[m->ops[i].t_in[k]]++;
        if (m->ops[i].t_out >= 0) writers[m->ops[i].t_out]++;
    }
    for (int i = 0; i + 1 < m->n_ops; i++) {
        mars_op_t *c = &m->ops[i], *s = &m->ops[i + 1];
        if (c->kind != OP_CONV_I8 || s->kind != OP_LUT_I8 || c->lut_off != NO_OFF || s->lut_off == NO_OFF) continue;
        const int q1 = c->t_out, q2 = s->t_out;
        if (q1 < 0 || q2 < 0 || s->t_in[0] != q1 || q1 == q2) continue;
        int inplace = 0; /* redirecting the result onto one of the convolution's own inputs would make a parallel launch run in place */
        for (int k = 0; k < c->n_in; k++)
            if (c->t_in[k] == q2) inplace = 1;
        if (inplace) continue;
        if (readers[q1] != 1 || writers[q1] != 1 || writers[q2] != 1 || m->mt[q1].io_out || m->mt[q1].io_in || m->mt[q1].is_weight || m->mt[q2].is_weight) continue;
        const size_t n1 = (size_t)c->out_h * c->out_w * c->out_c;
        if (s->n != n1 || m->mt[q2].bytes < n1) continue; /* the layer must cover exactly the convolution's result */
        if (m->pub.tensors[q1].desc.dtype != MARS_DTYPE_INT8 || m->pub.tensors[q2].desc.dtype != MARS_DTYPE_INT8) continue;
        const int8_t *tab = (const int8_t *)(m->arena_host + s->lut_off);
        c->lut_off = s->lut_off;
        if (mhip_conv_i8_lut2_ok(c->cs)) { /* the half-step form, as fuse_silu builds it (the ReLU clamp folds in) */
            int8_t tab2[512];
            for (int k = -256; k < 256; k++) {
                int r = k >= 0 ? (k + 1) >> 1 : -((1 - k) >> 1);
                const int lo = c->relu ? 0 : -128;
                r = r < lo ? lo : (r > 127 ? 127 : r);
                tab2[k + 256] = tab[r + 128];
            }
            c->lut2_off = arena_reserve(m, 512);
            if (c->lut2_off != NO_OFF) memcpy(m->arena_host + c->lut2_off, tab2, 512);
            c = &m->ops[i]; /* (arena_reserve may move the host image, not the ops) */
        }
        c->t_out = q2;
        touch(m, q2, n1);
        m->mt[q1].needed = 0;
        memmove(&m->ops[i + 1], &m->ops[i + 2], (size_t)(m->n_ops - i - 2) * sizeof(mars_op_t));
        m->n_ops -= 1;
    }
    free(readers);
    free(writers);
}

/* float32 form of the same chain: conv_f32 -> SIGMOID (float, :742-749) -> MUL (float, :807-816).  The epilogue evaluates
 * s = 1.0f / (1.0f + expf(-v)), out = v * s with the reference's roundings and this image's libm expf (expf_exact.h), so
 * the fused result is the same float bit for bit; the two intermediates are never written (5 of the 7 float passes over
 * the tensor disappear: the element-wise layers were 36 % of the float32 graph's time). */
void fuse_silu_f32(mars_model_ext_t *m) {
    const int nt = (int)m->pub.header.num_tensors;
    int *readers = (int *)calloc((size_t)nt + 1, sizeof(int));
    int *writers = (int *)calloc((size_t)nt + 1, sizeof(int));
    if (!readers || !writers) { free(readers); free(writers); return; }
    for (int i = 0; i < m->n_ops; i++) {
        for (int k = 0; k < m->ops[i].n_in; k++)
            if (m->ops[i].t_in[k] >= 0) readers[m->ops[i].t_in[k]]++;
        if (m->ops[i].t_out >= 0) writers[m->ops[i].t_out]++;
    }
    for (int i = 0; i + 2 < m->n_ops; i++) {
        mars_op_t *c = &m->ops[i], *s = &m->ops[i + 1], *mu = &m->ops[i + 2];
        if (c->kind != OP_CONV_F32 || s->kind != OP_SIGMOID_F32 || mu->kind != OP_BINARY_F32 || !mu->is_mul) continue;
        const int q1 = c->t_out, q2 = s->t_out, q3 = mu->t_out;
        if (s->t_in[0] != q1 || q1 == q2 || q2 == q3 || q1 == q3 || q3 == c->t_in[0]) continue;
        if (!((mu->t_in[0] == q1 && mu->t_in[1] == q2) || (mu->t_in[0] == q2 && mu->t_in[1] == q1))) continue;
        if (readers[q1] != 2 || readers[q2] != 1 || writers[q1] != 1 || writers[q2] != 1 || writers[q3] != 1) continue;
        if (m->mt[q1].io_out || m->mt[q2].io_out || m->mt[q1].io_in || m->mt[q2].io_in || m->mt[q3].is_weight) continue;
        const size_t n1 = (size_t)c->out_h * c->out_w * c->out_c;
        if (s->n != n1 || mu->n != n1) continue; /* the chain must cover exactly the conv's result */
        c->silu_f32 = 1;
        c->t_out = q3;
        touch(m, q3, n1 * 4);
        m->mt[q1].needed = 0;
        m->mt[q2].needed = 0;
        memmove(&m->ops[i + 1], &m->ops[i + 3], (size_t)(m->n_ops - i - 3) * sizeof(mars_op_t));
        m->n_ops -= 2;
    }
    free(readers);
    free(writers);
}

/* ------------------------------------------------------------- zero-copy concat
 * A concat input that (a) has exactly one producer launch of a kind that can write a channel slice
 * (int8 NHWC conv, int8 add/mul, max-pool, upsample), (b) is read by nothing but that concat, and
 * (c) has the concat's pixel grid, is produced directly inside the concat's output tensor: the
 * producer gets (pixel stride = concat channels, channel offset) and the copy launch disappears.
 * Same bytes end up in the concat output; the intermediate tensor is never materialised. */
void elide_concat(mars_model_ext_t *m) {
    const int nt = (int)m->pub.header.num_tensors;
    int *readers = (int *)calloc((size_t)nt + 1, sizeof(int));
    int *writers = (int *)calloc((size_t)nt + 1, sizeof(int));
    if (!readers || !writers) { free(readers); free(writers); return; }
    for (int i = 0; i < m->n_ops; i++) {
        for (int k = 0; k < m->ops[i].n_in; k++)
            if (m->ops[i].t_in[k] >= 0) readers[m->ops[i].t_in[k]]++;
        if (m->ops[i].t_out >= 0) writers[m->ops[i].t_out]++;
    }
    for (int i = 0; i < m->n_ops; i++) {
        mars_op_t *cs = &m->ops[i];
        if (cs->kind != OP_CONCAT_SLICE) continue;
        const int ti = cs->t_in[0], to = cs->t_out;
        const mtensor_t *mt = &m->mt[ti];
        if (mt->is_weight || mt->io_in || mt->io_out || readers[ti] != 1 || writers[ti] != 1) continue;
        int j = -1;
        for (int k = 0; k < i; k++)
            if (m->ops[k].t_out == ti) j = k;
        if (j < 0) continue;
        mars_op_t *pr = &m->ops[j];
        if (pr->out_pix_stride || pr->add_t) continue; /* a folded Add needs its other operand laid out like the output */
        const size_t npix = (size_t)cs->out_h * cs->out_w;
        int ok = 0;
        if (pr->kind == OP_CONV_I8 && !pr->nchw && !pr->out_nchw) ok = (size_t)pr->out_h * pr->out_w == npix && pr->out_c == cs->in_c;
        else if (pr->kind == OP_BINARY_I8) ok = pr->n == npix * (size_t)cs->in_c;
        else if (pr->kind == OP_MAXPOOL || pr->kind == OP_UPSAMPLE) ok = (size_t)pr->out_h * pr->out_w == npix && pr->in_c == cs->in_c;
        if (!ok || cs->ch_off + cs->in_c > cs->out_c) continue;
        /* nothing between producer and the copy may touch the concat output except its other slices */
        int clash = 0;
        for (int k = j; k < i && !clash; k++) {
            const mars_op_t *o = &m->ops[k];
            if (o->kind == -1) continue; /* a slice copy already elided */
            for (int q = 0; q < o->n_in; q++)
                if (o->t_in[q] == to) clash = 1;
            if (o->t_out == to && o->kind != OP_CONCAT_SLICE && !o->out_pix_stride) clash = 1;
        }
        if (clash) continue;
        pr->t_out = to;
        pr->out_pix_stride = cs->out_c;
        pr->out_ch_off = cs->ch_off;
        if (pr->kind == OP_BINARY_I8) pr->in_c = cs->in_c; /* channel run of the slice */
        pr->bytes += 0;
        m->mt[ti].needed = 0;
        cs->kind = -1; /* dropped below */
    }
    int w = 0;
    for (int i = 0; i < m->n_ops; i++)
        if (m->ops[i].kind != -1) m->ops[w++] = m->ops[i];
    m->n_ops = w;
    free(readers);
    free(writers);
}

/* Dead stores of a CONCAT.  The reference copies input k's "pixels" (runs of shape[3] BYTES, whatever the dtype) to byte
 * p * out_c + off_k of the output, one input after the other (mars_runtime.c:977-996).  When shape[3] of an input EQUALS the
 * output's (the NCHW-tagged float twins: every tensor's shape[3] is the map width), a slice is one contiguous run of
 * npix * out_c bytes starting at off_k, and the next input's run, starting out_c bytes later, overwrites all of it but its first
 * off_j - off_k bytes before anything can read them.  Such a slice copies only the pixels that hold surviving bytes (the partly
 * surviving last one whole: the later slice, launched after it on the same stream, rewrites the rest) -- the tensor's final bytes
 * are the reference's.  On the float yolov5 twins this halves the concat traffic (2.7 -> 1.4 ms per batch of 256). */
void trim_concat(mars_model_ext_t *m) {
    for (int i = 0; i < m->n_ops; i++) {
        mars_op_t *k = &m->ops[i];
        if (k->kind != OP_CONCAT_SLICE || k->in_c != k->out_c || k->out_c <= 0) continue;
        const size_t npix = (size_t)k->out_h * k->out_w;
        for (int q = i + 1; q < m->n_ops && m->ops[q].layer == k->layer; q++) {
            const mars_op_t *j = &m->ops[q];
            if (j->kind != OP_CONCAT_SLICE || j->t_out != k->t_out || j->in_c != j->out_c || j->out_c != k->out_c) continue;
            if ((size_t)j->out_h * j->out_w != npix || j->ch_off < k->ch_off) continue;
            const size_t live = (size_t)(j->ch_off - k->ch_off), live_pix = (live + (size_t)k->out_c - 1) / (size_t)k->out_c;
            if (live_pix < (size_t)k->out_h * k->out_w) {
                k->out_h = 1;
                k->out_w = (int)live_pix; /* 0: nothing of this slice survives (the launcher skips an empty copy) */
                k->bytes = 2.0 * (double)live_pix * k->in_c;
            }
        }
    }
}

/* NCHW-tagged int8 graphs (every shipped model: SURVEY a7, mxu_conv.c:630-670) on the NHWC kernels without a relayout per layer (round 6).
 * conv2d_int8_mxu reads [C][H][W] and writes [O][H][W]; the matrix-core kernels want pixels x channels, so every such convolution used to
 * relay its input into scratch (nchw_to_nhwc_kernel: 44 % of yolov5n_int8.mars' GPU time at batch 256) and stage its result through LDS
 * for the planar store -- and no fusion pass took them.  But a tensor that only convolutions and ELEMENT-WISE layers (LUT maps, Mul / Add:
 * functions of corresponding bytes, whatever the order) touch need not hold the reference's byte order at all: it is kept as
 * [H * W][C] on the device (mtensor_t.nhwc_c), the convolution reading it skips the relayout (op->nchw = 0), the one writing it stores
 * rows (op->out_nchw = 0), and the passes that follow (Add folding, paired launches) see plain NHWC convolutions.  Same values at the
 * same (channel, pixel): the reference's NCHW and NHWC kernels compute the same sums (mxu_conv.c:639-667 against :722-754).  Tensors a
 * byte-wise layer touches (MAXPOOL / CONCAT / UPSAMPLE index bytes by shape[1..3] whatever the tag: :919-1041), graph inputs and
 * outputs keep the reference's bytes.  mars_hip_read_tensor / write_tensor convert; fusion level 0 keeps every tensor as tagged. */
void nhwc_internal(mars_model_ext_t *m) {
    const int nt = (int)m->pub.header.num_tensors;
    for (int i = 0; i < nt; i++) m->mt[i].nhwc_c = m->mt[i].nhwc_hw = m->mt[i].nhwc_pitch = m->mt[i].partial = 0;
    if (m->fusion < 1 || getenv("MARS_HIP_NO_NHWC_INTERNAL")) return;
    unsigned char *el = (unsigned char *)calloc((size_t)nt + 1, 1);
    if (!el) return;
    for (int t = 0; t < nt; t++) { /* candidates: dense [1, C, H, W] int8 activations, C a multiple of 16 (16-byte pixel rows), not NHWC-tagged */
        const mars_tensor_t *d = &m->pub.tensors[t].desc;
        const mtensor_t *mt = &m->mt[t];
        if (mt->is_weight || mt->io_in || mt->io_out || !mt->needed || d->dtype != MARS_DTYPE_INT8 || d->format == MARS_FORMAT_NHWC || d->ndims != 4 ||
            d->shape[0] != 1 || d->shape[1] <= 0 || (d->shape[1] & 15) || d->shape[2] <= 0 || d->shape[3] <= 0)
            continue;
        if (mt->bytes != (size_t)d->shape[1] * d->shape[2] * d->shape[3]) continue; /* (what every op touches of it is checked per op below) */
        el[t] = 1;
    }
    /* CONCAT layers whose inputs and output all have the same H, W (every C3 concat of the shipped files): the reference's byte logic is
     * then a fixed function of flat indices (move.hip: concat_nchwq_kernel), computable on pixels x channels operands -- the layer's
     * slices become ONE launch (OP_CONCAT_Q) if its tensors all stay eligible, else they all keep the reference's bytes.  grp[i] = index of
     * the first slice op of op i's layer when that layer qualifies, else -1 */
    int *grp = (int *)malloc(sizeof(int) * (size_t)(m->n_ops + 1));
    if (!grp) { free(el); return; }
    for (int i = 0; i < m->n_ops; i++) grp[i] = -1;
    for (int i = 0; i < m->n_ops; i++) {
        const mars_op_t *o = &m->ops[i];
        if (o->kind != OP_CONCAT_SLICE || grp[i] != -1 || (i > 0 && m->ops[i - 1].kind == OP_CONCAT_SLICE && m->ops[i - 1].layer == o->layer)) continue;
        int n = 0, ok = 1;
        while (i + n < m->n_ops && m->ops[i + n].kind == OP_CONCAT_SLICE && m->ops[i + n].layer == o->layer) n++;
        const mars_layer_t *Ld = &m->pub.layers[o->layer].desc;
        const mars_tensor_t *od = o->t_out >= 0 ? &m->pub.tensors[o->t_out].desc : NULL;
        if (!od || n < 1 || n > 4 || (uint32_t)n != Ld->num_inputs || od->ndims != 4) ok = 0;
        for (int k = 0; k < n && ok; k++) {
            const mars_op_t *q = &m->ops[i + k];
            const mars_tensor_t *id = q->t_in[0] >= 0 ? &m->pub.tensors[q->t_in[0]].desc : NULL;
            /* slice k: runs of W bytes at offset k W, over C_out x H "pixels" (shape[1] x shape[2] of the output) */
            if (!id || id->ndims != 4 || q->t_out != o->t_out || q->t_in[0] == o->t_out || q->out_pix_stride || id->shape[2] != od->shape[2] ||
                id->shape[3] != od->shape[3] || q->in_c != od->shape[3] || q->out_c != od->shape[3] || q->ch_off != k * od->shape[3] ||
                q->out_h != od->shape[1] || q->out_w != od->shape[2] || (long)od->shape[1] * od->shape[2] * od->shape[3] > 0x7fffffffL)
                ok = 0;
            for (int k2 = 0; k2 < k && ok; k2++)
                if (m->ops[i + k2].t_in[0] == q->t_in[0]) ok = 0; /* (one tensor twice: keep it simple) */
        }
        if (ok)
            for (int k = 0; k < n; k++) grp[i + k] = i;
    }
#define DIMS_ARE(t, c, h, w) (m->pub.tensors[t].desc.shape[1] == (c) && m->pub.tensors[t].desc.shape[2] == (h) && m->pub.tensors[t].desc.shape[3] == (w))
    for (int pass = 0, changed = 1; changed && pass < nt + 2; pass++) {
        changed = 0;
        for (int i = 0; i < m->n_ops; i++) {
            const mars_op_t *o = &m->ops[i];
            if (o->kind == OP_CONCAT_SLICE && grp[i] >= 0) { /* a qualifying CONCAT layer: all of its tensors, or none */
                if (grp[i] != i) continue;
                int all = el[o->t_out];
                for (int k = i; k < m->n_ops && grp[k] == i; k++) all = all && el[m->ops[k].t_in[0]];
                if (!all) {
                    if (el[o->t_out]) { el[o->t_out] = 0; changed = 1; }
                    for (int k = i; k < m->n_ops && grp[k] == i; k++)
                        if (el[m->ops[k].t_in[0]]) { el[m->ops[k].t_in[0]] = 0; changed = 1; }
                }
                continue;
            }
            if ((o->kind == OP_UPSAMPLE || (o->kind == OP_MAXPOOL && !o->chain_n)) && o->t_in[0] >= 0 && o->t_out >= 0 && o->t_in[0] != o->t_out &&
                !o->out_pix_stride && !o->out_ch_off) {
                /* the byte-wise UPSAMPLE / stride-1 MAXPOOL as fixed functions of flat indices (move.hip *_nchwq_kernel): both tensors or neither */
                const mars_tensor_t *id = &m->pub.tensors[o->t_in[0]].desc, *od = &m->pub.tensors[o->t_out].desc;
                int fits = id->ndims == 4 && od->ndims == 4 && o->in_h == id->shape[1] && o->in_w == id->shape[2] && o->in_c == id->shape[3] &&
                           o->out_h == od->shape[1] && o->out_w == od->shape[2] && (long)o->out_h * o->out_w * o->in_c <= (long)od->shape[1] * od->shape[2] * od->shape[3];
                if (o->kind == OP_MAXPOOL)
                    fits = fits && o->sh == 1 && o->sw == 1 && o->kh >= 1 && o->kh <= 17 && o->kw >= 1 && od->shape[1] == id->shape[1] &&
                           od->shape[2] == id->shape[2] && od->shape[3] == id->shape[3];
                if (fits && el[o->t_in[0]] && el[o->t_out]) continue;
                /* (else: the generic path below takes both tensors out) */
            }
            int ts[8], n = 0, ok[8];
            for (int k = 0; k < o->n_in && k < 4; k++) { ts[n] = o->t_in[k]; ok[n++] = 0; }
            const int n_in = n;
            if (o->t_out >= 0) { ts[n] = o->t_out; ok[n++] = 0; }
            if (o->kind == OP_CONV_I8 && !o->nseg && !o->add_t && !o->pre && !o->out_pix_stride && !o->chain_n) {
                /* input side: the tagged-NCHW view of a tensor of exactly these dims; output side likewise */
                if (n_in == 1 && o->nchw && ts[0] >= 0 && el[ts[0]] && DIMS_ARE(ts[0], o->in_c, o->in_h, o->in_w) && o->c_pad == o->in_c) ok[0] = 1;
                if (o->t_out >= 0 && o->out_nchw && el[o->t_out] && DIMS_ARE(o->t_out, o->out_c, o->out_h, o->out_w) && ts[0] != o->t_out) ok[n - 1] = 1;
            } else if ((o->kind == OP_LUT_I8 || (o->kind == OP_BINARY_I8 && !o->out_pix_stride)) && o->t_out >= 0) {
                /* element-wise: every operand and the result in the same order, or none of them */
                int all = 1;
                for (int k = 0; k < n; k++)
                    if (ts[k] < 0 || !el[ts[k]] || o->n != m->mt[ts[k]].bytes || !DIMS_ARE(ts[k], m->pub.tensors[ts[0]].desc.shape[1], m->pub.tensors[ts[0]].desc.shape[2], m->pub.tensors[ts[0]].desc.shape[3]))
                        all = 0;
                for (int k = 0; k < n; k++) ok[k] = all;
            }
            for (int k = 0; k < n; k++)
                if (ts[k] >= 0 && el[ts[k]] && !ok[k]) { el[ts[k]] = 0; changed = 1; }
            for (int k = 0; k < o->chain_n && k < 3; k++)
                if (o->chain_out[k] >= 0 && el[o->chain_out[k]]) { el[o->chain_out[k]] = 0; changed = 1; }
            for (int k = 0; k < o->nseg && k < 4; k++)
                if (o->seg_t[k] >= 0 && el[o->seg_t[k]]) { el[o->seg_t[k]] = 0; changed = 1; }
        }
    }
#undef DIMS_ARE
    for (int i = 0; i < m->n_ops; i++) {
        mars_op_t *o = &m->ops[i];
        if (o->kind == OP_CONCAT_SLICE && grp[i] == i && el[o->t_out]) { /* the layer's slices -> one launch on pixels x channels operands */
            const mars_tensor_t *od = &m->pub.tensors[o->t_out].desc;
            int n = 0;
            double bytes = (double)m->mt[o->t_out].bytes;
            for (int k = i; k < m->n_ops && grp[k] == i; k++, n++) {
                o->t_in[n] = m->ops[k].t_in[0];
                bytes += (double)m->mt[m->ops[k].t_in[0]].bytes;
                if (k != i) m->ops[k].kind = -1;
            }
            o->kind = OP_CONCAT_Q;
            o->n_in = n;
            o->out_c = od->shape[1]; o->in_h = od->shape[2]; o->in_w = od->shape[3];
            o->out_h = od->shape[2]; o->out_w = od->shape[3];
            o->bytes = bytes;
            continue;
        }
        if ((o->kind == OP_UPSAMPLE || (o->kind == OP_MAXPOOL && !o->chain_n)) && o->t_in[0] >= 0 && o->t_out >= 0 && el[o->t_in[0]] && el[o->t_out]) {
            o->kind = o->kind == OP_UPSAMPLE ? OP_UPSAMPLE_Q : OP_MAXPOOL_Q; /* (the fixpoint left both tensors eligible only where the op fits) */
            continue;
        }
        if (o->kind != OP_CONV_I8) continue;
        if (o->n_in >= 1 && o->t_in[0] >= 0 && el[o->t_in[0]]) o->nchw = 0;
        if (o->t_out >= 0 && el[o->t_out]) o->out_nchw = 0;
    }
    {
        int w = 0;
        for (int i = 0; i < m->n_ops; i++)
            if (m->ops[i].kind != -1) m->ops[w++] = m->ops[i];
        m->n_ops = w;
    }
    free(grp);
    for (int t = 0; t < nt; t++)
        if (el[t]) {
            m->mt[t].nhwc_c = m->pub.tensors[t].desc.shape[1];
            m->mt[t].nhwc_hw = m->pub.tensors[t].desc.shape[2] * m->pub.tensors[t].desc.shape[3];
            /* every layer left on it touches exactly its nominal bytes (checked above); what the replaced concat slices would have read
             * past them (their runs cover the OUTPUT's byte count) no longer counts -- with it a concat input had another frame stride than
             * its producer's other operand and the residual Add in front of a C3's concat was not folded */
            m->mt[t].extent = m->mt[t].bytes;
        }
    /* Write-only convolution results with a ragged channel count (the 255-channel Detect convolutions of the shipped files: their readers are
     * no-op layers): nothing in the graph reads them, so they too are kept pixels x channels, at a 16-byte-aligned pixel pitch (255 -> 256; the
     * pad channel has zero weights, its byte is never read) -- the convolution then takes the aligned row epilogue and the tile walkers instead
     * of the planar store through LDS.  mars_hip_read_tensor un-pads and converts. */
    for (int i = 0; i < m->n_ops; i++) {
        mars_op_t *o = &m->ops[i];
        const int T = o->t_out;
        if (o->kind != OP_CONV_I8 || !o->out_nchw || T < 0 || el[T] || (o->out_c & 15) == 0 || (o->in_c & 15) || o->out_pix_stride || o->out_ch_off ||
            o->add_t || o->nseg || o->pre)
            continue;
        const mars_tensor_t *d = &m->pub.tensors[T].desc;
        const mtensor_t *mt = &m->mt[T];
        const int P = (o->out_c + 15) & ~15;
        if (mt->is_weight || mt->io_in || mt->io_out || d->dtype != MARS_DTYPE_INT8 || d->format == MARS_FORMAT_NHWC || d->ndims != 4 || d->shape[0] != 1 ||
            d->shape[1] != o->out_c || d->shape[2] != o->out_h || d->shape[3] != o->out_w || P > o->oc_pad ||
            mt->bytes != (size_t)o->out_c * o->out_h * o->out_w || mt->extent != mt->bytes)
            continue;
        int touched = 0; /* any other op that reads or writes it keeps it as tagged */
        for (int j = 0; j < m->n_ops && !touched; j++) {
            const mars_op_t *q = &m->ops[j];
            if (j != i && q->t_out == T) touched = 1;
            for (int k = 0; k < q->n_in && k < 4; k++)
                if (q->t_in[k] == T) touched = 1;
            for (int k = 0; k < q->nseg && k < 4; k++)
                if (q->seg_t[k] == T) touched = 1;
            for (int k = 0; k < q->chain_n && k < 3; k++)
                if (q->chain_out[k] == T) touched = 1;
        }
        if (touched) continue;
        o->out_nchw = 0;
        o->out_pix_stride = P;
        o->store_c = P;
        m->mt[T].nhwc_c = o->out_c;
        m->mt[T].nhwc_hw = o->out_h * o->out_w;
        m->mt[T].nhwc_pitch = P;
        touch(m, T, (size_t)o->out_h * o->out_w * P);
    }
    free(el);
}

static int op_writes(const mars_op_t *o, int t);

/* The reference's CONCAT on NCHW-tagged maps of equal size is a shift of its LAST input by N - 1 map rows (move.hip: concat_nchwq_kernel):
 *   out(c, h, w) = in_last(c, h - (N - 1), w)  for h >= N - 1 and c < C_last,   0 for c >= C_last;   the first N - 1 rows are irregular.
 * When its only reader is a plain 1 x 1 convolution X (C3's cv3, SPPF's cv2 in the shipped files), X over rows h >= N - 1 is the SAME
 * convolution over in_last with its K loop cut to the C_last channels that are not zero, its output N - 1 rows further down:
 *   - the concat produces its first N - 1 rows only (rows_only; the tensor becomes `partial`),
 *   - X is split in two launches: rows 0 .. N - 2 from those rows (all channels, the original weights), rows N - 1 .. H - 1 straight from
 *     in_last (a second weight image: the first C_last input channels).
 * The concat's copy of the whole map and the zero half (or three quarters) of X's K loop go.  Same bytes out: every output row is the sum
 * the reference computes, over exactly the non-zero terms.  Runs last of the fusion passes. */
void virtual_concat_q(mars_model_ext_t *m) {
    if (m->fusion < 1 || getenv("MARS_HIP_NO_VCONCAT_Q")) return;
    for (int i = 0; i < m->n_ops; i++) {
        mars_op_t *cq = &m->ops[i];
        if (cq->kind != OP_CONCAT_Q || cq->n_in < 2 || cq->rows_only) continue;
        const int T = cq->t_out, N = cq->n_in, H = cq->in_h, W = cq->in_w, Cout = cq->out_c;
        const int TL = cq->t_in[N - 1];
        const int CL = m->mt[TL].nhwc_c;
        if (H < N + 1 || CL <= 0 || (CL & 15) || CL >= Cout || m->mt[TL].nhwc_pitch || m->mt[T].nhwc_pitch) continue;
        int r = -1, nr = 0, bad = 0;
        for (int j = 0; j < m->n_ops; j++) {
            const mars_op_t *o = &m->ops[j];
            for (int k = 0; k < o->n_in && k < 4; k++)
                if (o->t_in[k] == T) { nr++; if (r < 0) r = j; }
            for (int k = 0; k < o->nseg && k < 4; k++)
                if (o->seg_t[k] == T) bad = 1;
            if (j != i && op_writes(o, T)) bad = 1;
        }
        if (bad || r <= i || nr < 1 || nr > 2) continue;
        /* one reader, or the two of a paired launch (a head C3's cv1 + cv2: both are split, both halves stay pairs) */
        const int np = m->ops[r].pair_next ? 2 : 1;
        if (nr != np || (r > 0 && m->ops[r - 1].pair_next) || (np == 2 && (r + 1 >= m->n_ops || m->ops[r + 1].t_in[0] != T || m->ops[r + 1].pair_next))) continue;
        int ok = 1;
        for (int q = 0; q < np && ok; q++) {
            const mars_op_t *cv = &m->ops[r + q];
            if (cv->kind != OP_CONV_I8 || cv->kh != 1 || cv->kw != 1 || cv->sh != 1 || cv->sw != 1 || cv->pt || cv->pl || cv->nchw || cv->out_nchw ||
                cv->nseg || cv->add_t || cv->pre || cv->out_pix_stride || cv->out_ch_off ||
                cv->n_in != 1 || cv->in_c != Cout || cv->in_h != H || cv->in_w != W || cv->out_h != H || cv->out_w != W || cv->in_byte_off ||
                cv->out_byte_off || cv->t_out == TL || cv->t_out == T || (cv->out_c & 15) || m->mt[cv->t_out].nhwc_pitch)
                ok = 0;
        }
        if (!ok) continue;
        int clash = 0; /* in_last must still hold at the readers what it held at the concat */
        for (int j = i + 1; j < r + np && !clash; j++)
            if (op_writes(&m->ops[j], TL)) clash = 1;
        if (clash) continue;
        /* the second weight images: input channels [0, C_last) of every output channel (1 x 1: OIHW and OHWI coincide) */
        size_t w_off2[2] = {NO_OFF, NO_OFF};
        int row_pad2 = 0;
        for (int q = 0; q < np && ok; q++) {
            const mars_op_t *cv = &m->ops[r + q];
            const mars_layer_t *L = &m->pub.layers[cv->layer].desc;
            const int tw = find_tensor(m, L->params.conv.weight_tensor_id);
            int oc_pad2, c_eff2;
            mhip_conv_i8_pack_geom(CL, 1, cv->out_c, &row_pad2, &oc_pad2, &c_eff2);
            if (tw < 0 || oc_pad2 != cv->oc_pad || c_eff2 != CL) { ok = 0; break; }
            const size_t k64 = ALIGN_UP((size_t)row_pad2, 64);
            w_off2[q] = arena_reserve(m, (size_t)oc_pad2 * k64);
            if (w_off2[q] == NO_OFF) return;
            cv = &m->ops[r + q];
            if (!m->deferred) {
                const size_t full = (size_t)cv->out_c * Cout, cut = (size_t)cv->out_c * CL;
                int8_t *tmp = (int8_t *)malloc(full ? full : 1), *sel = (int8_t *)malloc(cut ? cut : 1);
                if (!tmp || !sel) { free(tmp); free(sel); m->plan_err = MARS_ERR_ALLOC_FAILED; return; }
                blob_read(m, (size_t)m->pub.tensors[tw].desc.data_offset, full, tmp);
                for (int oc = 0; oc < cv->out_c; oc++) memcpy(sel + (size_t)oc * CL, tmp + (size_t)oc * Cout, (size_t)CL);
                mars_pack_conv_i8(sel, cut, 1, cv->out_c, CL, 1, 1, c_eff2, row_pad2, oc_pad2, (int8_t *)m->arena_host + w_off2[q]);
                free(tmp); free(sel);
            }
        }
        if (!ok) continue; /* (an image reserved for the first of a pair stays unused: a few KB of the arena) */
        /* ops: [concat: first N - 1 rows] ... [the readers over those rows] [the readers over in_last] */
        while (m->n_ops + np > m->cap_ops) {
            const int cap = m->cap_ops * 2;
            mars_op_t *npp = (mars_op_t *)realloc(m->ops, (size_t)cap * sizeof(mars_op_t));
            if (!npp) { m->plan_err = MARS_ERR_ALLOC_FAILED; return; }
            m->ops = npp; m->cap_ops = cap;
        }
        cq = &m->ops[i];
        memmove(&m->ops[r + 2 * np], &m->ops[r + np], sizeof(mars_op_t) * (size_t)(m->n_ops - r - np));
        m->n_ops += np;
        for (int q = np - 1; q >= 0; q--) m->ops[r + np + q] = m->ops[r + q]; /* (np == 2: [a b] -> [a b a b]) */
        for (int q = 0; q < np; q++) {
            mars_op_t *top = &m->ops[r + q], *mainop = &m->ops[r + np + q];
            top->in_h = top->out_h = N - 1;
            top->macs = top->macs * (N - 1) / H;
            top->bytes = (double)(N - 1) * W * (Cout + top->out_c);
            top->w2_off = NO_OFF; top->w2_rows = 0;
            mainop->t_in[0] = TL;
            mainop->in_c = CL; mainop->c_pad = CL; mainop->row_pad = row_pad2;
            mainop->in_h = mainop->out_h = H - (N - 1);
            mainop->w_off = w_off2[q];
            mainop->w2_off = NO_OFF; mainop->w2_rows = 0;
            mainop->out_byte_off = (size_t)(N - 1) * W * (size_t)mainop->out_c;
            mainop->macs = (double)(H - (N - 1)) * W * mainop->out_c * CL;
            mainop->bytes = (double)(H - (N - 1)) * W * (CL + mainop->out_c);
            mainop->variant = 0;
        }
        cq->rows_only = N - 1;
        cq->bytes = (double)(N - 1) * W * Cout * 2.0;
        m->mt[T].partial = 1;
    }
}

/* Residual Add folded into the convolution that produces one of its operands (the bottleneck shortcut of C3):
 * out = Add(conv_result, x) is evaluated in the convolution's epilogue with the reference's float steps
 * (mars_runtime.c ADD branch: (a*sa + b*sb) * (1/so) + 0.5f, truncated, saturated), reading x where the output
 * goes.  Saves writing the convolution result and reading it back.  Conditions: the convolution result has no
 * other reader, x and the Add output have the same dense layout and frame stride, nothing touches them in
 * between, and the scales keep the float -> int conversion in range (so no x86 fix-up is needed). */
static size_t planned_stride(const mtensor_t *t) {
    size_t s = ALIGN_UP(t->extent > t->bytes ? t->extent : t->bytes, 256);
    return s ? s : 256;
}
void fuse_add(mars_model_ext_t *m) {
    const int nt = (int)m->pub.header.num_tensors;
    int *readers = (int *)calloc((size_t)nt + 1, sizeof(int));
    int *writers = (int *)calloc((size_t)nt + 1, sizeof(int));
    if (!readers || !writers) { free(readers); free(writers); return; }
    for (int i = 0; i < m->n_ops; i++) {
        for (int k = 0; k < m->ops[i].n_in; k++)
            if (m->ops[i].t_in[k] >= 0) readers[m->ops[i].t_in[k]]++;
        if (m->ops[i].t_out >= 0) writers[m->ops[i].t_out]++;
    }
    for (int j = 0; j < m->n_ops; j++) {
        mars_op_t *ad = &m->ops[j];
        if (ad->kind != OP_BINARY_I8 || ad->is_mul || ad->n_in != 2 || ad->out_pix_stride) continue;
        for (int side = 0; side < 2; side++) {
            const int A = ad->t_in[side], X = ad->t_in[1 - side], O = ad->t_out;
            if (A < 0 || X < 0 || O < 0 || A == X || O == X || O == A) continue;
            if (readers[A] != 1 || writers[A] != 1 || m->mt[A].io_in || m->mt[A].io_out || m->mt[A].is_weight) continue;
            if (m->mt[X].is_weight || m->mt[O].is_weight || writers[O] != 1) continue;
            int i = -1;
            for (int k = 0; k < j; k++)
                if (m->ops[k].t_out == A) i = k;
            if (i < 0) continue;
            mars_op_t *c = &m->ops[i];
            if (c->kind != OP_CONV_I8 || c->out_nchw || !c->safe || c->out_pix_stride || (c->out_c & 15) || (c->in_c & 15) ||
                c->nseg || c->add_t || c->n_in != 1)
                continue;
            if (ad->n != (size_t)c->out_h * c->out_w * c->out_c) continue;
            if (c->t_in[0] == O) continue; /* add(conv(X), Y) -> X: sequential in the reference, a race when fused */
            if (planned_stride(&m->mt[X]) != planned_stride(&m->mt[O])) continue;
            const float s_conv = side == 0 ? ad->f0 : ad->f1, s_other = side == 0 ? ad->f1 : ad->f0, inv = ad->f2;
            const double bound = 128.0 * (fabs((double)s_conv) + fabs((double)s_other)) * fabs((double)inv) + 1.0;
            if (!(bound < 2147483000.0)) continue; /* also rejects NaN / inf */
            int clash = 0;
            for (int k = i; k <= j && !clash; k++) {
                const mars_op_t *o = &m->ops[k];
                if (o->t_out == X) clash = 1; /* x must be complete before the convolution runs */
                if (k > i && k < j) {
                    if (o->t_out == O) clash = 1;
                    for (int q = 0; q < o->n_in; q++)
                        if (o->t_in[q] == O) clash = 1;
                }
            }
            if (clash) continue;
            c->t_out = O;
            c->add_t = X + 1;
            c->add_s_conv = s_conv; c->add_s_other = s_other; c->add_inv = inv;
            c->t_in[c->n_in++] = X;
            c->bytes += (double)ad->n;
            m->mt[A].needed = 0;
            ad->kind = -1;
            readers[A] = 0;
            break;
        }
    }
    int w = 0;
    for (int i = 0; i < m->n_ops; i++)
        if (m->ops[i].kind != -1) m->ops[w++] = m->ops[i];
    m->n_ops = w;
    free(readers);
    free(writers);
}

/* The float32 form: Add(conv_f32 result, x) (reference mars_runtime.c:807-816: out[i] = a[i] + b[i]) evaluated where the
 * convolution stores -- one float add behind the (fused) SiLU, the same float the separate layer computes from the same two
 * floats, so every kernel form stays what it was (the exact-order kernel bit-identical).  Saves a write and two reads of
 * the tensor per bottleneck (7 launches, 7 % of the float32 twin's step). */
void fuse_add_f32(mars_model_ext_t *m) {
    const int nt = (int)m->pub.header.num_tensors;
    int *readers = (int *)calloc((size_t)nt + 1, sizeof(int));
    int *writers = (int *)calloc((size_t)nt + 1, sizeof(int));
    if (!readers || !writers) { free(readers); free(writers); return; }
    for (int i = 0; i < m->n_ops; i++) {
        for (int k = 0; k < m->ops[i].n_in; k++)
            if (m->ops[i].t_in[k] >= 0) readers[m->ops[i].t_in[k]]++;
        if (m->ops[i].t_out >= 0) writers[m->ops[i].t_out]++;
    }
    for (int j = 0; j < m->n_ops; j++) {
        mars_op_t *ad = &m->ops[j];
        if (ad->kind != OP_BINARY_F32 || ad->is_mul || ad->n_in != 2) continue;
        for (int side = 0; side < 2; side++) {
            const int A = ad->t_in[side], X = ad->t_in[1 - side], O = ad->t_out;
            if (A < 0 || X < 0 || O < 0 || A == X || O == X || O == A) continue;
            if (readers[A] != 1 || writers[A] != 1 || m->mt[A].io_in || m->mt[A].io_out || m->mt[A].is_weight) continue;
            if (m->mt[X].is_weight || m->mt[O].is_weight || writers[O] != 1) continue;
            int i = -1;
            for (int k = 0; k < j; k++)
                if (m->ops[k].t_out == A) i = k;
            if (i < 0) continue;
            mars_op_t *c = &m->ops[i];
            if (c->kind != OP_CONV_F32 || c->add_t || c->n_in != 1) continue;
            if (ad->n != (size_t)c->out_h * c->out_w * c->out_c) continue;
            if (c->t_in[0] == O) continue; /* add(conv(X), Y) -> X: sequential in the reference, a race when fused */
            int clash = 0;
            for (int k = i; k <= j && !clash; k++) {
                const mars_op_t *o = &m->ops[k];
                if (o->t_out == X) clash = 1; /* x must be complete before the convolution runs */
                if (k > i && k < j) {
                    if (o->t_out == O) clash = 1;
                    for (int q = 0; q < o->n_in; q++)
                        if (o->t_in[q] == O) clash = 1;
                }
            }
            if (clash) continue;
            c->t_out = O;
            c->add_t = X + 1;
            c->t_in[c->n_in++] = X;
            c->bytes += 4.0 * (double)ad->n;
            m->mt[A].needed = 0;
            ad->kind = -1;
            readers[A] = 0;
            break;
        }
    }
    int w = 0;
    for (int i = 0; i < m->n_ops; i++)
        if (m->ops[i].kind != -1) m->ops[w++] = m->ops[i];
    m->n_ops = w;
    free(readers);
    free(writers);
}

/* Fused C3 bottleneck: conv1x1 + SiLU (A) whose only reader is the k x k convolution B right behind it (B usually
 * carries the folded residual Add of A's input) -> B evaluates A on its staged input patch (conv_i8_patch<PRE>); A's
 * output tensor is never written.  Same bytes: B sees, at every in-image pixel of its window, exactly the int8 value A
 * would have stored there, and zeros outside the image as its SAME padding prescribes.  Only where the device code can
 * take it (mhip_conv_i8_pre_ok: stride 1, 32 / 64 channels, patch fits); everything else keeps the two launches.
 * Fusion level 2 only: measured on the yolov5s twin it removes 4 launches (batch 1: 0.494 -> 0.485 ms) but returns
 * nothing at batch 256 -- these 32 / 64-channel layers are bound by the requantisation's vector instructions, not by
 * the bytes the fusion saves, and the halo makes the fused kernel requantise 1.3x the pixels. */
void fuse_bottleneck(mars_model_ext_t *m) {
    const int nt = (int)m->pub.header.num_tensors;
    int *readers = (int *)calloc((size_t)nt + 1, sizeof(int));
    if (!readers) return;
    for (int i = 0; i < m->n_ops; i++)
        for (int k = 0; k < m->ops[i].n_in; k++)
            if (m->ops[i].t_in[k] >= 0) readers[m->ops[i].t_in[k]]++;
    for (int i = 0; i + 1 < m->n_ops; i++) {
        mars_op_t *a = &m->ops[i], *b = &m->ops[i + 1];
        if (a->kind != OP_CONV_I8 || b->kind != OP_CONV_I8 || a->pre || b->pre) continue;
        const int T = a->t_out;
        if (T < 0 || b->t_in[0] != T || readers[T] != 1 || m->mt[T].io_in || m->mt[T].io_out || m->mt[T].is_weight) continue;
        if (a->kh != 1 || a->kw != 1 || a->sh != 1 || a->sw != 1 || a->nchw || b->nchw || a->out_nchw || b->out_nchw || !a->safe || !b->safe || a->nseg || b->nseg ||
            a->add_t || a->n_in != 1 || a->out_pix_stride || a->relu || a->lut2_off == NO_OFF || b->lut2_off == NO_OFF ||
            a->in_c != a->out_c || a->out_c != b->in_c || (a->in_c != 32 && a->in_c != 64) || b->sh != 1 || b->sw != 1 ||
            a->in_h != b->in_h || a->in_w != b->in_w || a->b_off == NO_OFF || a->pair_next || b->pair_next ||
            (i > 0 && m->ops[i - 1].pair_next))
            continue;
        const int X = a->t_in[0];
        if (X < 0 || X == b->t_out || m->mt[X].is_weight) continue;
        /* the device side decides on geometry: describe B with A folded in (pointers only need to be non-null here) */
        mars_op_t trial = *b;
        trial.pre = 1;
        trial.t_in[0] = X;
        mhip_conv_i8_t p;
        memset(&p, 0, sizeof(p));
        p.frames = 1; p.in_c = trial.in_c; p.in_h = trial.in_h; p.in_w = trial.in_w; p.out_h = trial.out_h; p.out_w = trial.out_w;
        p.out_c = trial.store_c ? trial.store_c : trial.out_c; p.kh = trial.kh; p.kw = trial.kw; p.stride_h = trial.sh; p.stride_w = trial.sw;
        p.pad_top = trial.pt; p.pad_left = trial.pl; p.row_pad = trial.row_pad; p.oc_pad = trial.oc_pad; p.safe = trial.safe;
        p.out_pix_stride = trial.out_pix_stride; p.out_ch_off = trial.out_ch_off;
        p.pre_w = (const int8_t *)m; p.pre_bias = (const int32_t *)m; p.pre_lut2 = (const uint8_t *)m; p.lut2 = (const uint8_t *)m;
        p.lut = (const uint8_t *)m;
        if (!mhip_conv_i8_pre_ok(&p)) continue;
        b->pre = 1;
        b->pre_w_off = a->w_off; b->pre_b_off = a->b_off; b->pre_lut2_off = a->lut2_off; b->pre_cs = a->cs;
        b->t_in[0] = X;
        b->macs += a->macs;
        m->mt[T].needed = 0;
        readers[T] = 0;
        a->kind = -1;
    }
    int w = 0;
    for (int i = 0; i < m->n_ops; i++)
        if (m->ops[i].kind != -1) m->ops[w++] = m->ops[i];
    m->n_ops = w;
    free(readers);
}

/* Virtual concat: when every reader of a Concat output is a plain 1x1 convolution, the concat tensor is never
 * written -- the convolution's K loop takes each run of channels straight from the tensor that owns it
 * (conv_i8_persist<SEG>).  The slice copies disappear and every producer keeps writing dense rows.  Same bytes
 * as the reference: the copy would have put pixel p, channels [off, off+in_c) of the concat tensor = pixel p of the
 * input, which is exactly what the segmented read fetches (checked: nothing rewrites an input between the copy's
 * position and the last reader). */
void virtual_concat(mars_model_ext_t *m) {
    const int nt = (int)m->pub.header.num_tensors;
    for (int T = 0; T < nt; T++) {
        const mtensor_t *mt = &m->mt[T];
        if (mt->is_weight || mt->io_in || mt->io_out) continue;
        int sl[4], ns = 0, bad = 0, first_reader = -1, last_reader = -1, last_slice = -1;
        for (int i = 0; i < m->n_ops && !bad; i++) {
            const mars_op_t *o = &m->ops[i];
            if (o->t_out == T) {
                if (o->kind != OP_CONCAT_SLICE || ns >= 4) bad = 1;
                else { sl[ns++] = i; last_slice = i; }
            }
            for (int k = 0; k < o->n_in; k++)
                if (o->t_in[k] == T) {
                    if (o->kind != OP_CONV_I8 || k != 0 || o->n_in != 1 || o->nchw || o->kh != 1 || o->kw != 1 || o->sh != 1 ||
                        o->sw != 1 || o->pt || o->pl || !o->safe || o->nseg || (o->out_c & 15) || o->in_h != o->out_h ||
                        o->in_w != o->out_w || (o->in_c & (o->in_c - 1)) != 0 || o->add_t) /* the tile walker: K position by
                                                                                            * shifts, no folded Add */
                        bad = 1;
                    if (first_reader < 0) first_reader = i;
                    last_reader = i;
                }
        }
        if (bad || ns < 2 || first_reader < 0 || last_slice > first_reader) continue;
        /* slices in channel order, tiling [0, out_c) in multiples of 32, all over the same pixels */
        for (int a = 0; a < ns; a++)
            for (int b = a + 1; b < ns; b++)
                if (m->ops[sl[b]].ch_off < m->ops[sl[a]].ch_off) { int t = sl[a]; sl[a] = sl[b]; sl[b] = t; }
        int c = 0;
        const mars_op_t *s0 = &m->ops[sl[0]];
        for (int a = 0; a < ns && !bad; a++) {
            const mars_op_t *o = &m->ops[sl[a]];
            if (o->ch_off != c || (o->in_c & 31) || o->out_h != s0->out_h || o->out_w != s0->out_w || o->out_c != s0->out_c ||
                o->t_in[0] == T || o->t_in[0] < 0)
                bad = 1;
            c += o->in_c;
            /* the input must still hold at the last reader what it held where the copy stood */
            for (int i = sl[a] + 1; i <= last_reader && !bad; i++)
                if (m->ops[i].t_out == o->t_in[0]) bad = 1;
        }
        if (bad || c != s0->out_c) continue;
        for (int i = 0; i < m->n_ops && !bad; i++) {
            const mars_op_t *o = &m->ops[i];
            if (o->kind == OP_CONV_I8 && o->n_in == 1 && o->t_in[0] == T &&
                (o->in_c != c || (size_t)o->in_h * o->in_w != (size_t)s0->out_h * s0->out_w))
                bad = 1;
        }
        if (bad) continue;
        for (int i = 0; i < m->n_ops; i++) {
            mars_op_t *o = &m->ops[i];
            if (o->kind != OP_CONV_I8 || o->n_in != 1 || o->t_in[0] != T) continue;
            o->nseg = ns;
            o->n_in = ns;
            for (int a = 0; a < ns; a++) {
                o->seg_t[a] = o->t_in[a] = m->ops[sl[a]].t_in[0];
                o->seg_c[a] = m->ops[sl[a]].in_c;
            }
        }
        for (int a = 0; a < ns; a++) m->ops[sl[a]].kind = -1;
        m->mt[T].needed = 0;
        /* a segment that is a 2x2 nearest upsample (reference :1003-1044) of a half-size tensor, read by nothing
         * else: the convolution reads the half-size tensor at pixel (y/2, x/2) and the upsample launch goes too */
        for (int a = 0; a < ns; a++) {
            const int S = m->ops[sl[a]].t_in[0];
            int up = -1, nread = 0, nwrite = 0, ok = 1;
            for (int i = 0; i < m->n_ops; i++) {
                const mars_op_t *o = &m->ops[i];
                if (o->kind == -1) continue;
                if (o->t_out == S) { nwrite++; up = i; }
                for (int k = 0; k < o->n_in; k++)
                    if (o->t_in[k] == S && !(o->kind == OP_CONV_I8 && o->nseg == ns && o->seg_t[a] == S)) nread++;
            }
            if (nwrite != 1 || nread != 0 || up < 0 || m->mt[S].io_out || m->mt[S].io_in || m->mt[S].is_weight) continue;
            const mars_op_t *u = &m->ops[up];
            if (u->kind != OP_UPSAMPLE || u->out_pix_stride || u->scale_h != 2 || u->scale_w != 2 || u->out_h != 2 * u->in_h ||
                u->out_w != 2 * u->in_w || u->in_c != m->ops[sl[a]].in_c || u->out_h != s0->out_h || u->out_w != s0->out_w)
                continue;
            const int U = u->t_in[0];
            if (U < 0 || U == S) continue;
            int last = -1;
            for (int i = 0; i < m->n_ops; i++)
                if (m->ops[i].kind == OP_CONV_I8 && m->ops[i].nseg == ns && m->ops[i].seg_t[a] == S) last = i;
            for (int i = up; i <= last && ok; i++)
                if (m->ops[i].kind != -1 && m->ops[i].t_out == U) ok = 0; /* the half-size tensor must stay as it was */
            if (!ok) continue;
            for (int i = 0; i < m->n_ops; i++) {
                mars_op_t *o = &m->ops[i];
                if (o->kind == OP_CONV_I8 && o->nseg == ns && o->seg_t[a] == S) {
                    o->seg_t[a] = o->t_in[a] = U;
                    o->seg_up |= 1 << a;
                }
            }
            m->ops[up].kind = -1;
            m->mt[S].needed = 0;
        }
    }
    int w = 0;
    for (int i = 0; i < m->n_ops; i++)
        if (m->ops[i].kind != -1) m->ops[w++] = m->ops[i];
    m->n_ops = w;
}

/* Two convolutions that read the same input with the same geometry (C3's cv1 and cv2, which the exporter emits
 * a few layers apart) are launched as ONE grid (conv_i8_persist<PAIR>): the workgroups that need a pixel tile run
 * next to each other, so the input is read from HBM once.  The later one is moved up behind the earlier one when
 * nothing in between touches its operands or its output. */
static int same_conv_input(const mars_op_t *a, const mars_op_t *b) {
    if (a->n_in != b->n_in || a->nseg != b->nseg || a->seg_up != b->seg_up) return 0;
    for (int k = 0; k < a->n_in; k++)
        if (a->t_in[k] != b->t_in[k]) return 0;
    for (int k = 0; k < a->nseg; k++)
        if (a->seg_t[k] != b->seg_t[k] || a->seg_c[k] != b->seg_c[k]) return 0;
    return a->in_h == b->in_h && a->in_w == b->in_w && a->in_c == b->in_c && a->out_h == b->out_h && a->out_w == b->out_w &&
           a->kh == b->kh && a->kw == b->kw && a->sh == b->sh && a->sw == b->sw && a->pt == b->pt && a->pl == b->pl &&
           a->row_pad == b->row_pad && a->oc_pad == b->oc_pad;
}
static int pairable(const mars_op_t *o) {
    /* measured: pairs with a plain input gain 10-20 %, pairs reading a virtual concat lose (their single launches are
     * tuned individually), so only the former are formed */
    return o->kind == OP_CONV_I8 && !o->nchw && !o->out_nchw && !o->add_t && !o->nseg && o->safe && o->lut_off != NO_OFF && !o->pair_next &&
           (o->in_c & 15) == 0 && o->in_c > 4 && (o->out_c & 15) == 0 && !o->out_pix_stride;
}
static int op_writes(const mars_op_t *o, int t) {
    if (o->t_out == t) return 1;
    for (int k = 0; k < o->chain_n; k++)
        if (o->chain_out[k] == t) return 1;
    return 0;
}
void pair_convs(mars_model_ext_t *m) {
    for (int i = 0; i + 1 < m->n_ops; i++) {
        mars_op_t *a = &m->ops[i];
        if (!pairable(a) || (i > 0 && m->ops[i - 1].pair_next)) continue;
        for (int j = i + 1; j < m->n_ops && j <= i + 48; j++) {
            mars_op_t *b = &m->ops[j];
            if (!pairable(b) || !same_conv_input(a, b) || b->t_out == a->t_out) continue;
            int ok = 1;
            for (int k = 0; k < b->n_in; k++)
                if (b->t_in[k] == b->t_out || b->t_in[k] == a->t_out) ok = 0;
            for (int q = i + 1; q < j && ok; q++) {
                const mars_op_t *o = &m->ops[q];
                if (op_writes(o, b->t_out)) ok = 0;
                for (int k = 0; k < o->n_in; k++)
                    if (o->t_in[k] == b->t_out) ok = 0;
                for (int k = 0; k < b->n_in; k++)
                    if (op_writes(o, b->t_in[k])) ok = 0;
            }
            if (!ok) continue;
            mars_op_t moved = *b;
            memmove(&m->ops[i + 2], &m->ops[i + 1], sizeof(mars_op_t) * (size_t)(j - i - 1));
            m->ops[i + 1] = moved;
            m->ops[i].pair_next = 1;
            break;
        }
    }
}

/* The float twins' C3: cv1 and cv2 are two 1 x 1 convolutions over the same tensor with the same shape.  As one grid (conv_f32_split's
 * pair form, mhip_conv_f32_pair) the second one's input reads hit L2: at float32 the input of these layers is 4 bytes per element and
 * they are bound by exactly those bytes.  Formed where both take conv_f32_split (a weight image packed under f32_mfma = 3 / 4 at load),
 * neither has a folded Add; the second is moved up next to the first when nothing in between touches its tensors (as pair_convs). */
static int pairable_f32(const mars_op_t *o) {
    return o->kind == OP_CONV_F32 && !o->add_t && o->n_in == 1 && o->w2_off != NO_OFF && o->w3_off == NO_OFF && !o->pair_next && !o->in_rec && !o->out_rec &&
           o->kh == 1 && o->kw == 1;
}
void pair_convs_f32(mars_model_ext_t *m) {
    if (getenv("MARS_HIP_NO_PAIR_F32")) return;
    for (int i = 0; i + 1 < m->n_ops; i++) {
        mars_op_t *a = &m->ops[i];
        if (!pairable_f32(a) || (i > 0 && m->ops[i - 1].pair_next)) continue;
        for (int j = i + 1; j < m->n_ops && j <= i + 48; j++) {
            mars_op_t *b = &m->ops[j];
            if (!pairable_f32(b) || b->t_in[0] != a->t_in[0] || b->t_out == a->t_out || b->t_out < 0 || a->t_out < 0) continue;
            if (a->in_h != b->in_h || a->in_w != b->in_w || a->in_c != b->in_c || a->out_h != b->out_h || a->out_w != b->out_w || a->out_c != b->out_c ||
                a->sh != b->sh || a->sw != b->sw || a->pt != b->pt || a->pl != b->pl || a->silu_f32 != b->silu_f32 || a->f32_exact != b->f32_exact)
                continue;
            if (planned_stride(&m->mt[a->t_out]) != planned_stride(&m->mt[b->t_out])) continue;
            if (a->w2_planes != b->w2_planes) continue;
            int ok = b->t_in[0] != b->t_out && b->t_in[0] != a->t_out;
            for (int q = i + 1; q < j && ok; q++) {
                const mars_op_t *o = &m->ops[q];
                if (op_writes(o, b->t_out) || op_writes(o, b->t_in[0])) ok = 0;
                for (int k = 0; k < o->n_in; k++)
                    if (o->t_in[k] == b->t_out) ok = 0;
                if (o->pair_next) ok = 0; /* (a pair stays adjacent) */
            }
            if (!ok) continue;
            mars_op_t moved = *b;
            memmove(&m->ops[i + 2], &m->ops[i + 1], sizeof(mars_op_t) * (size_t)(j - i - 1));
            m->ops[i + 1] = moved;
            m->ops[i].pair_next = 1;
            break;
        }
    }
}

/* SPPF: MaxPool -> MaxPool -> MaxPool, stride 1, same window, each feeding the next: one launch that keeps the
 * frame in LDS (mhip_pool_chain_i8).  Every stage's tensor is still written (the concat / convolution reads them). */
void fuse_pool_chains(mars_model_ext_t *m) {
    for (int i = 0; i + 1 < m->n_ops; i++) {
        mars_op_t *a = &m->ops[i];
        if (a->kind != OP_MAXPOOL || a->chain_n || a->sh != 1 || a->sw != 1 || a->out_pix_stride || (a->in_c & 15) ||
            a->out_h != a->in_h || a->out_w != a->in_w || a->kh <= 0 || a->kw <= 0 || (size_t)a->in_h * a->in_w * 64 > 60 * 1024 ||
            a->t_out < 0 || m->mt[a->t_out].io_out)
            continue;
        int n = 1;
        a->chain_out[0] = a->t_out;
        while (n < 3 && i + n < m->n_ops) {
            const mars_op_t *b = &m->ops[i + n];
            if (b->kind != OP_MAXPOOL || b->t_in[0] != a->chain_out[n - 1] || b->sh != 1 || b->sw != 1 || b->out_pix_stride ||
                b->in_c != a->in_c || b->in_h != a->in_h || b->in_w != a->in_w || b->out_h != a->in_h || b->out_w != a->in_w ||
                b->kh != a->kh || b->kw != a->kw || b->t_out < 0 || m->mt[b->t_out].io_out || b->t_out == a->t_in[0])
                break;
            a->chain_out[n] = b->t_out;
            n++;
        }
        if (n < 2) continue;
        a->chain_n = n;
        for (int k = 1; k < n; k++) {
            a->bytes += m->ops[i + k].bytes - (double)a->in_h * a->in_w * a->in_c; /* later stages re-read nothing */
            m->ops[i + k].kind = -1;
        }
        a->t_out = a->chain_out[n - 1];
    }
    int w = 0;
    for (int i = 0; i < m->n_ops; i++)
        if (m->ops[i].kind != -1) m->ops[w++] = m->ops[i];
    m->n_ops = w;
}

/* float32 graphs: which convolutions may take the f32 matrix cores (fused rounding per tap, inside the 1e-4 tolerance)?
 * The reference's MAXPOOL runs int8 byte logic on whatever bytes it is given (mars_runtime.c:919-957), and so does the
 * fused-ReLU clamp of a float convolution (:700-707): over float bytes both are DISCONTINUOUS functions of their input (a
 * last-bit change can flip which byte wins / whether a mantissa byte is zeroed: measured, 1 value in 16 384 left the
 * tolerance), so every convolution from which one of them can be reached keeps the reference's summation order (conv_f32_kernel, bit-identical); byte-copying
 * layers (concat, upsample) and the float element-wise layers only pass small differences on. */
/* Record-format tensors between float convolutions (f32_mfma = 3 at load, fusion >= 1).  conv_f32_patch cuts every input value into
 * two bf16 pieces and lays them out channels-last in LDS -- per K step ~130 vector instructions of its staging phase, which is what
 * bounds it (profiles/r05_experiments.md).  Where the tensor a k x k convolution X reads is written by another convolution P and read
 * by nothing else (a C3 bottleneck's 1 x 1 -> 3 x 3, the stem -> layer 3, a C3's last 1 x 1 -> the stride-2 convolution behind it), P
 * writes those pieces itself ([c / 8][h][w] records of 32 bytes, the same bytes per element: mhip_conv_f32_t.out_rec) and X fills its
 * patch ring by LDS-DMA (conv_f32_prec).  Same arithmetic: X multiplies exactly the pieces it would have cut itself.  The tensor's
 * bytes are then not the reference's floats: mars_hip_read_tensor / write_tensor convert (hi + mid, the value to 2^-16).  Depends on
 * shapes and the mode only (descriptor-only ranks decide alike). */
void rec_pairs(mars_model_ext_t *m) {
    const int nt = (int)m->pub.header.num_tensors;
    for (int i = 0; i < nt; i++) m->mt[i].rec_c = m->mt[i].rec_hw = 0;
    m->rec_skipped = 0;
    m->rec_max_frames = (size_t)-1;
    if (m->fusion < 1 || getenv("MARS_HIP_NO_REC") || mhip_conv_f32_mode(-1) != 3) return;
    const char *lim_env = getenv("MARS_HIP_REC_LIMIT"); /* (tests lower it to see the per-batch decision with small tensors) */
    const size_t frames = m->rec_frames > 0 ? (size_t)m->rec_frames : 1, lim = lim_env ? (size_t)strtoull(lim_env, NULL, 0) : (size_t)0xfffffff0u; /* 32-bit byte offsets over all frames of a tensor */
    int *readers = (int *)calloc((size_t)nt + 1, sizeof(int));
    int *writers = (int *)calloc((size_t)nt + 1, sizeof(int));
    if (!readers || !writers) { free(readers); free(writers); return; }
    for (int i = 0; i < m->n_ops; i++) {
        const mars_op_t *o = &m->ops[i];
        for (int k = 0; k < o->n_in && k < 4; k++)
            if (o->t_in[k] >= 0) readers[o->t_in[k]]++;
        /* (the operand of a folded Add is one of t_in[]: fuse_add_f32) */
        for (int k = 0; k < o->nseg && k < 4; k++)
            if (o->seg_t[k] >= 0) readers[o->seg_t[k]]++;
        if (o->t_out >= 0) writers[o->t_out]++;
        for (int k = 0; k < o->chain_n && k < 3; k++)
            if (o->chain_out[k] >= 0) writers[o->chain_out[k]]++;
    }
    for (int j = 0; j < m->n_ops; j++) {
        mars_op_t *x = &m->ops[j];
        if (x->kind != OP_CONV_F32 || x->w3_off == NO_OFF || x->w3_stem || x->n_in != (x->add_t ? 2 : 1) || x->in_rec) continue;
        const int T = x->t_in[0];
        if (T < 0 || readers[T] != 1 || writers[T] != 1 || m->mt[T].io_in || m->mt[T].io_out || m->mt[T].is_weight) continue;
        if (x->add_t && x->add_t - 1 == T) continue;
        int i = -1;
        for (int k = 0; k < j; k++)
            if (m->ops[k].t_out == T) i = k;
        if (i < 0) continue;
        mars_op_t *pr = &m->ops[i];
        if (pr->kind != OP_CONV_F32 || pr->add_t || pr->out_rec || (pr->out_c & 7) || pr->kh > 32 || pr->kw > 32) continue;
        if (pr->pair_next || (i > 0 && m->ops[i - 1].pair_next)) continue; /* (a paired launch writes plain floats) */
        if (pr->out_c != x->in_c || pr->out_h != x->in_h || pr->out_w != x->in_w) continue;
        if (!(pr->w2_off != NO_OFF || (pr->w3_off != NO_OFF && pr->w3_stem))) continue; /* conv_f32_split or conv_f32_stem writes it */
        /* a weight image alone does not mean conv_f32_split takes the shape (stride 2 with an odd kernel width and pad > 1, maps
         * narrower than a gather ...): it would decline at launch, and nothing else writes records -- the run would fail */
        if (pr->w2_off != NO_OFF && mhip_conv_f32_split_takes(pr->out_c, pr->in_c, pr->kh, pr->kw, pr->sh, pr->sw, pr->pl, pr->in_w, pr->out_w) < 0) continue;
        /* ... and the reader declines a folded Add whose operand has another frame stride than its output (conv_f32_try_patch) */
        if (x->add_t && planned_stride(&m->mt[x->add_t - 1]) != planned_stride(&m->mt[x->t_out])) continue;
        if (m->mt[T].bytes != (size_t)x->in_c * x->in_h * x->in_w * 4) continue;
        const int form = mhip_conv_f32_patch_rec_form(x->out_c, x->in_c, x->kh, x->kw, x->sw, x->pl, x->in_h, x->in_w, x->out_h, x->out_w);
        if (!form) continue;
        if (form == 1) { /* conv_f32_prec: its image replaces the plain one in place (same size by construction) */
            const size_t n0 = mhip_conv_f32_patch_pack2(x->out_c, x->in_c, x->kh, x->kw, x->sw, x->pl, x->in_h, x->in_w, x->out_h, x->out_w, 0, NULL, NULL);
            const size_t n1 = mhip_conv_f32_patch_pack2(x->out_c, x->in_c, x->kh, x->kw, x->sw, x->pl, x->in_h, x->in_w, x->out_h, x->out_w, 1, NULL, NULL);
            if (!n1 || n1 != n0) continue;
        }
        { /* the three tensors the two launches address with 32-bit offsets: all frames of each must stay below 4 GiB */
            const int tt[3] = {pr->t_in[0], T, x->t_out};
            size_t worst = 0;
            for (int q = 0; q < 3; q++)
                if (tt[q] >= 0) {
                    const mtensor_t *u = &m->mt[tt[q]];
                    const size_t st = ALIGN_UP(u->extent > u->bytes ? u->extent : u->bytes, 256);
                    if (st > worst) worst = st;
                }
            if (worst && worst * frames > lim) {
                m->rec_skipped = 1;
                continue;
            }
            if (worst && lim / worst < m->rec_max_frames) m->rec_max_frames = lim / worst;
        }
        if (form == 1 && !m->deferred)
            mhip_conv_f32_patch_pack2(x->out_c, x->in_c, x->kh, x->kw, x->sw, x->pl, x->in_h, x->in_w, x->out_h, x->out_w, 1,
                                      (const float *)(m->arena_host + x->w_off), m->arena_host + x->w3_off);
        x->in_rec = form;
        pr->out_rec = 1;
        m->mt[T].rec_c = x->in_c;
        m->mt[T].rec_hw = x->in_h * x->in_w;
    }
    free(readers);
    free(writers);
}

/* Exact zeros the reference's byte-wise CONCAT leaves in float tensors (round 6).  The concat copies runs of shape[3] BYTES whatever the dtype
 * (mars_runtime.c:971-999): on [1, C, H, W] float tensors it writes C H W bytes (+ the slices' offsets) of an output that holds 4 C H W -- in
 * the private zero-initialised buffers of the parity target everything behind that is zero in every frame, i.e. all but the
 * first C / 4 + 1 channels (measured on the twins: tests).  A 1 x 1 convolution that reads such a tensor (every C3's cv3, SPPF's cv2, the head
 * C3s' cv1 + cv2) multiplies zeros for three quarters of its K loop.  Under the split-bf16 modes (3 / 4: results inside the tolerance, not
 * bit-equal anyway) its K loop stops at the last channel that can be non-zero (mhip_conv_f32_t.k_limit): acc + (+-0 * w) == acc for finite w,
 * so the sums are the same floats (up to the sign of a zero); skipped weights are checked to be finite where the blob is at hand.  Modes 0 - 2
 * keep the full loop.  Depends on shapes only. */
void zero_tail_f32(mars_model_ext_t *m) {
    const int nt = (int)m->pub.header.num_tensors;
    for (int i = 0; i < nt; i++) m->mt[i].zero_from = 0;
    for (int i = 0; i < m->n_ops; i++) m->ops[i].k_limit = 0;
    if (m->fusion < 1 || getenv("MARS_HIP_NO_ZERO_TAIL")) return;
    for (int j = 0; j < m->n_ops; j++) {
        mars_op_t *x = &m->ops[j];
        if (x->kind != OP_CONV_F32 || x->kh != 1 || x->kw != 1 || x->sh != 1 || x->sw != 1 || x->pt || x->pl || x->in_rec || x->w2_off == NO_OFF) continue;
        const int T = x->t_in[0];
        if (T < 0 || m->mt[T].is_weight || m->mt[T].io_in || m->pub.tensors[T].desc.dtype != MARS_DTYPE_FLOAT32) continue;
        if (m->mt[T].bytes != (size_t)x->in_c * x->in_h * x->in_w * 4) continue;
        size_t written = 0; /* bytes [0, written) may be non-zero: the furthest byte any writer reaches */
        int ok = 1, nw = 0;
        for (int q = 0; q < m->n_ops && ok; q++) {
            const mars_op_t *o = &m->ops[q];
            for (int k = 0; k < o->chain_n && k < 3; k++)
                if (o->chain_out[k] == T) ok = 0;
            if (o->t_out != T) continue;
            if (o->kind != OP_CONCAT_SLICE || o->out_pix_stride) { ok = 0; break; }
            nw++;
            const size_t npix = (size_t)o->out_h * o->out_w;
            if (npix && o->in_c > 0) {
                const size_t end = (npix - 1) * (size_t)o->out_c + (size_t)o->ch_off + (size_t)o->in_c;
                if (end > written) written = end;
            }
        }
        if (!ok || !nw || !written) continue;
        const size_t plane = (size_t)x->in_h * x->in_w * 4;
        const size_t kl = (written + plane - 1) / plane; /* channels that hold a written byte */
        if (kl >= (size_t)x->in_c) continue;
        if (!m->deferred) { /* 0 * w must be 0: every skipped weight finite */
            const float *w = (const float *)(m->arena_host + x->w_off);
            int finite = 1;
            for (int oc = 0; oc < x->out_c && finite; oc++)
                for (int ic = (int)kl; ic < x->in_c; ic++) {
                    const float v = w[(size_t)oc * x->in_c + ic];
                    if (!(v - v == 0.0f)) { finite = 0; break; }
                }
            if (!finite) continue;
        }
        x->k_limit = (int)kl;
        if (!m->mt[T].zero_from || written > m->mt[T].zero_from) m->mt[T].zero_from = written;
        /* the layer's algorithmic work is what is not provably zero */
        x->macs = (double)x->out_h * x->out_w * x->out_c * (double)kl;
        x->bytes = ((double)x->in_h * x->in_w * (double)kl + (double)x->out_h * x->out_w * x->out_c) * 4.0;
    }
}

/* A float CONCAT that no launch materialises (round 6; conv_f32_vcat.hip has the arithmetic).  On N float maps [1, C, H, W] of equal size the reference's
 * byte-wise CONCAT (trim_concat has already cut it to what survives) leaves, per frame: the first W BYTES of every input but the last, then the first
 * L = C_out H W bytes of the LAST input shifted by sB = (N - 1) W bytes, then zeros.  Its readers -- one 1 x 1 convolution, or a pair of them --
 * already stop their K loop behind the written bytes (zero_tail_f32: k_limit = L / plane + 1).  For every pixel from sB / 4 on, such a convolution over
 * the concat IS the convolution over the address (last input - sB) with L / plane input planes: it runs on that view (vc_shift, k_limit required),
 * and a second, tiny launch (OP_CONV_F32_VHEAD) recomputes the first sB / 4 pixels of every output plane from their true operands.  The copies
 * (two passes over a quarter of the concat tensor) and the tensor itself go.  Same tolerance class as the launch it replaces (split-bf16 modes only).
 * Batch-dependent like rec_pairs (32-bit offsets over all frames): shares its per-batch re-planning.  Runs last. */
void virtual_concat_f32(mars_model_ext_t *m) {
    const int mode = mhip_conv_f32_mode(-1);
    if (m->fusion < 1 || getenv("MARS_HIP_NO_VCONCAT_F32") || (mode != 3 && mode != 4)) return;
    const char *lim_env = getenv("MARS_HIP_REC_LIMIT");
    const size_t frames = m->rec_frames > 0 ? (size_t)m->rec_frames : 1, lim = lim_env ? (size_t)strtoull(lim_env, NULL, 0) : (size_t)0xfffffff0u;
    for (int i = 0; i < m->n_ops; i++) {
        if (m->ops[i].kind != OP_CONCAT_SLICE) continue;
        const int T = m->ops[i].t_out, layer = m->ops[i].layer;
        int e = i;
        while (e + 1 < m->n_ops && m->ops[e + 1].kind == OP_CONCAT_SLICE && m->ops[e + 1].layer == layer && m->ops[e + 1].t_out == T) e++;
        const int N = e - i + 1, first_op = i;
        i = e; /* (whatever happens below, the scan goes on behind the group) */
        if (N < 2 || N > 4 || T < 0 || m->mt[T].io_in || m->mt[T].io_out || m->mt[T].is_weight || m->pub.tensors[T].desc.dtype != MARS_DTYPE_FLOAT32) continue;
        const mars_op_t *last = &m->ops[e];
        const int Wb = last->in_c; /* bytes per run */
        if (Wb <= 0 || (Wb & 3) || last->out_c != Wb || last->ch_off != (N - 1) * Wb || last->out_pix_stride || last->out_h <= 0 || last->out_w <= 0) continue;
        const int TL = last->t_in[0];
        const size_t Lb = (size_t)last->out_h * last->out_w * (size_t)Wb, sB = (size_t)(N - 1) * (size_t)Wb;
        if (TL < 0 || TL == T || m->mt[TL].is_weight || m->mt[TL].io_in /* (a pipeline may hand the graph input over in a buffer of its own: nothing mapped in front) */ || m->mt[TL].rec_c || m->mt[TL].nhwc_c || m->mt[TL].pix_stride || m->mt[TL].bytes < Lb ||
            m->pub.tensors[TL].desc.dtype != MARS_DTYPE_FLOAT32 || sB > 256)
            continue;
        int ok = 1, firsts[3] = {-1, -1, -1};
        for (int k = 0; k < N - 1 && ok; k++) {
            const mars_op_t *sl = &m->ops[first_op + k];
            const int ti = sl->t_in[0];
            if (sl->in_c != Wb || sl->out_c != Wb || sl->ch_off != k * Wb || sl->out_h != 1 || sl->out_w != 1 || sl->out_pix_stride || ti < 0 || ti == T ||
                m->mt[ti].is_weight || m->mt[ti].rec_c || m->mt[ti].nhwc_c || m->mt[ti].bytes < (size_t)Wb)
                ok = 0;
            firsts[k] = ti;
        }
        if (!ok) continue;
        /* who touches the concat tensor: the group writes it, one convolution (or one pair) reads it, nothing else */
        int r1 = -1, nr = 0;
        for (int j = 0; j < m->n_ops && ok; j++) {
            const mars_op_t *o = &m->ops[j];
            if (j >= first_op && j <= e) continue;
            if (op_writes(o, T)) ok = 0;
            for (int k = 0; k < o->nseg && k < 4; k++)
                if (o->seg_t[k] == T) ok = 0;
            for (int k = 0; k < o->n_in && k < 4; k++)
                if (o->t_in[k] == T) {
                    if (k != 0 || j <= e) ok = 0;
                    if (r1 < 0) r1 = j;
                    nr++;
                }
        }
        if (!ok || r1 < 0 || nr > 2) continue;
        const int paired = m->ops[r1].pair_next ? 1 : 0;
        if (nr != 1 + paired || (r1 > 0 && m->ops[r1 - 1].pair_next) || (paired && (r1 + 1 >= m->n_ops || m->ops[r1 + 1].t_in[0] != T))) continue;
        const int rl = r1 + paired; /* the last reader */
        size_t km = 0, worst = 0;
        for (int j = r1; j <= rl && ok; j++) {
            const mars_op_t *x = &m->ops[j];
            if (x->kind != OP_CONV_F32 || x->k_limit <= 0 || x->n_in != 1 || x->add_t || x->in_rec || x->vc_shift || x->kh != 1 || x->kw != 1 || x->sh != 1 || x->sw != 1 ||
                x->pt || x->pl || x->w2_off == NO_OFF || x->in_h != x->out_h || x->in_w != x->out_w || x->t_out == TL || x->t_out == T || x->t_out < 0) { ok = 0; break; }
            const size_t hw = (size_t)x->in_h * x->in_w, Pb = hw * 4;
            if (m->mt[T].bytes != (size_t)x->in_c * Pb || Lb % Pb || sB / 4 >= hw) { ok = 0; break; }
            km = Lb / Pb;
            if (km < 1 || (size_t)x->k_limit != km + 1 || km + 1 > (size_t)x->in_c) { ok = 0; break; }
            if (x->out_rec && (x->out_c & 7)) { ok = 0; break; }
            for (int k = 0; k < N - 1; k++)
                if (x->t_out == firsts[k]) ok = 0;
            const size_t st = planned_stride(&m->mt[x->t_out]);
            if (st > worst) worst = st;
        }
        if (!ok) continue;
        for (int j = e + 1; j <= rl && ok; j++) { /* the operands must still hold at the readers what they held at the concat */
            if (op_writes(&m->ops[j], TL)) ok = 0;
            for (int k = 0; k < N - 1; k++)
                if (op_writes(&m->ops[j], firsts[k])) ok = 0;
        }
        if (!ok) continue;
        if (planned_stride(&m->mt[TL]) > worst) worst = planned_stride(&m->mt[TL]);
        /* (the main launch's buffer range is the CONCAT tensor's size over the last input's frame stride: at most 3 strides more than frames of it) */
        if (worst * (frames + 3) > lim) { /* conv_f32_split addresses all frames of a tensor with 32-bit offsets and nothing else honours k_limit: as rec_pairs */
            m->rec_skipped = 1;
            continue;
        }
        if (lim / worst - 3 < m->rec_max_frames) m->rec_max_frames = lim / worst - 3;
        if (!m->deferred) { /* plane km is skipped for the pixels where the concat holds zeros: 0 * w must be 0 there (as zero_tail_f32 checks the planes behind it) */
            for (int j = r1; j <= rl && ok; j++) {
                const mars_op_t *x = &m->ops[j];
                const float *w = (const float *)(m->arena_host + x->w_off);
                for (int oc = 0; oc < x->out_c; oc++) {
                    const float v = w[(size_t)oc * x->in_c + km];
                    if (!(v - v == 0.0f)) { ok = 0; break; }
                }
            }
            if (!ok) continue;
        }
        /* ops: [slices: gone] ... [readers on the view] [one head launch per reader] */
        const int heads = 1 + paired;
        size_t wt_off[2] = {NO_OFF, NO_OFF}; /* the head launches' weights: planes 0 .. km transposed (mhip_conv_f32_vcat_pack) */
        for (int q = 0; q < heads; q++) {
            const mars_op_t *x = &m->ops[r1 + q];
            wt_off[q] = arena_reserve(m, mhip_conv_f32_vcat_pack(x->out_c, x->in_c, (int)km + 1, NULL, NULL));
            if (wt_off[q] == NO_OFF) return;
            if (!m->deferred)
                mhip_conv_f32_vcat_pack(x->out_c, x->in_c, (int)km + 1, (const float *)(m->arena_host + x->w_off), (float *)(m->arena_host + wt_off[q]));
        }
        while (m->n_ops + heads > m->cap_ops) {
            const int cap = m->cap_ops * 2;
            mars_op_t *np = (mars_op_t *)realloc(m->ops, (size_t)cap * sizeof(mars_op_t));
            if (!np) { m->plan_err = MARS_ERR_ALLOC_FAILED; return; }
            m->ops = np; m->cap_ops = cap;
        }
        memmove(&m->ops[rl + 1 + heads], &m->ops[rl + 1], sizeof(mars_op_t) * (size_t)(m->n_ops - rl - 1));
        m->n_ops += heads;
        for (int q = 0; q < heads; q++) {
            mars_op_t *x = &m->ops[r1 + q], *h = &m->ops[rl + 1 + q];
            const double hw = (double)x->in_h * x->in_w;
            x->t_in[0] = TL;
            x->vc_shift = (int)sB;
            x->k_limit = (int)km;
            x->macs = hw * x->out_c * (double)km;
            x->bytes = (hw * (double)km + hw * x->out_c) * 4.0;
            *h = *x;
            h->kind = OP_CONV_F32_VHEAD;
            h->pair_next = 0; h->vc_shift = 0;
            h->k_limit = (int)km + 1;
            h->vc_n = N - 1; h->vc_run = Wb / 4;
            h->n_in = N;
            for (int k = 0; k < N - 1; k++) { h->vc_t[k] = firsts[k]; h->t_in[1 + k] = firsts[k]; }
            h->macs = (double)(sB / 4) * x->out_c * (double)(km + 1);
            h->bytes = ((double)(sB / 4) * (double)(km + 1) + (double)(sB / 4) * x->out_c) * 4.0;
            h->variant = 0;
            h->w3_off = wt_off[q]; h->w3_stem = 0;
            h->w2_off = NO_OFF; h->w2_planes = 0;
        }
        memmove(&m->ops[first_op], &m->ops[e + 1], sizeof(mars_op_t) * (size_t)(m->n_ops - e - 1));
        m->n_ops -= N;
        m->mt[T].needed = 0;
        m->mt[T].partial = 1;
        m->mt[T].zero_from = 0;
        i = first_op - 1; /* (the array moved down by N: go on at the op that now stands where the group stood) */
    }
}

void f32_policy(mars_model_ext_t *m) {
    const int nt = (int)m->pub.header.num_tensors;
    unsigned char *hot = (unsigned char *)calloc((size_t)nt + 1, 1);
    if (!hot) {
        for (int i = 0; i < m->n_ops; i++) m->ops[i].f32_exact = 1;
        return;
    }
    for (int i = m->n_ops - 1; i >= 0; i--) {
        mars_op_t *o = &m->ops[i];
        int reach = 0;
        if (o->kind == OP_MAXPOOL && o->t_in[0] >= 0 && m->pub.tensors[o->t_in[0]].desc.dtype == MARS_DTYPE_FLOAT32) reach = 1;
        if (o->kind == OP_RELU_BYTES && o->t_out >= 0) { /* the fused-ReLU byte clamp over float bytes (:700-707): in place */
            hot[o->t_out] = 1;
            continue;
        }
        if (o->t_out >= 0 && hot[o->t_out]) reach = 1;
        for (int k = 0; k < o->chain_n; k++)
            if (o->chain_out[k] >= 0 && hot[o->chain_out[k]]) reach = 1;
        if (!reach) continue;
        for (int k = 0; k < o->n_in; k++)
            if (o->t_in[k] >= 0) hot[o->t_in[k]] = 1;
        if (o->kind == OP_CONV_F32) o->f32_exact = 1;
    }
    free(hot);
}

