/*
 * mars_preproc.c -- image front-end of the detection pipeline on the GPU: letterbox resize of uint8 RGB frames
 * to the model's input size and conversion to int8 (px - 128), written straight into the model's input tensor.
 *
 * Replaces reference src/mars/mars_yolo_test.c:40-77 (load_image(), after stbi_load): the reference resizes with
 * stbir_resize_uint8() of the stb_image_resize.h it vendors (include/stb/, third party), i.e. Catmull-Rom when an
 * axis grows, Mitchell when it shrinks or stays, clamped edges, linear colour space, float arithmetic.  The
 * per-axis filter tables depend only on (input size, output size); they are built here on the host with that
 * library's float steps (cited below) and turned into gather lists; csrc/hip/preproc.hip runs the two
 * accumulation passes with the library's operation order.  Bit-exact with the reference (tests/test_gpu_preproc.py).
 * There is no CPU pixel path: without the device every entry point fails.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../mhip.h"
#include "mars_hip.h"
#include "mars_internal.h"

/* ---- the two default kernels of the library (stb_image_resize.h:810-836), support 2 */
static float w_catmullrom(float d) {
    d = (float)fabs(d);
    if (d < 1.0f) return 1 - d * d * (2.5f - 1.5f * d);
    if (d < 2.0f) return 2 - d * (4 + d * (0.5f * d - 2.5f));
    return 0.0f;
}
static float w_mitchell(float d) {
    d = (float)fabs(d);
    if (d < 1.0f) return (16 + d * d * (21 * d - 36)) / 18;
    if (d < 2.0f) return (32 + d * (-60 + d * (36 - 7 * d))) / 18;
    return 0.0f;
}

#define TAPS 4 /* table row width of both kernels: ceil(support * 2), :900-906 */

typedef struct {
    int n_out, n_entries;
    int *start; /* [n_out + 1] */
    int *src;   /* [n_entries] clamped source index, increasing inside one output */
    float *w;   /* [n_entries] */
} gather_t;

static void gather_free(gather_t *g) {
    free(g->start);
    free(g->src);
    free(g->w);
    memset(g, 0, sizeof(*g));
}

static int clamp_index(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

/* rows[r*TAPS + t] is the library's flat coefficient table (rows may spill into the next row exactly as its
 * indexing does); first[r]..last[r] the window of row r.  `growing`: rows are outputs gathering inputs;
 * otherwise rows are inputs (offset by `margin`) scattering to outputs. */
static int build_axis(int in_n, int out_n, gather_t *g) {
    memset(g, 0, sizeof(*g));
    const float ratio = ((float)out_n / in_n) / (1.0f - 0.0f); /* :2221 with s0 = 0, s1 = 1 */
    const float shift = 0.0f * out_n / (1.0f - 0.0f);          /* :2224 */
    const int growing = ratio > 1;                             /* :864 */
    const int margin = growing ? 0 : (int)ceil(2.0f * 2 / ratio) / 2; /* :889, :897 */
    const int nrows = growing ? out_n : in_n + 2 * margin;            /* :908-914 */
    float *rows = (float *)calloc((size_t)nrows * TAPS + 64, sizeof(float));
    int *first = (int *)malloc(sizeof(int) * (size_t)nrows), *last = (int *)malloc(sizeof(int) * (size_t)nrows);
    g->start = (int *)calloc((size_t)out_n + 1, sizeof(int));
    int rc = -1;
    if (!rows || !first || !last || !g->start) goto done;
    for (int t = 0; t < 64; t++) rows[(size_t)nrows * TAPS + t] = 1.0f; /* stops the zero scan below; never weighted */

    if (growing) {
        const float reach = 2.0f * ratio; /* kernel support in output pixels, :1200 */
        for (int o = 0; o < out_n; o++) {
            const float oc = (float)o + 0.5f;
            const float lo = ((oc - reach) + shift) / ratio, hi = ((oc + reach) + shift) / ratio; /* :1010-1015 */
            const float at = (oc + shift) / ratio;                                                /* :1017 */
            int a = (int)floor(lo + 0.5), b = (int)floor(hi - 0.5);                               /* in double, :1018 */
            float *row = rows + (size_t)TAPS * o;
            float sum = 0;
            for (int t = 0; t <= b - a; t++) { /* :1049-1064 */
                row[t] = w_catmullrom(at - ((float)(t + a) + 0.5f));
                if (t == 0 && !row[t]) {
                    a++;
                    t--;
                    continue;
                }
                sum += row[t];
            }
            const float norm = 1 / sum; /* :1072 */
            for (int t = 0; t <= b - a; t++) row[t] *= norm;
            first[o] = a;
            last[o] = b;
            for (int t = b - a; t >= 0 && !row[t]; t--) last[o] = a + t - 1; /* :1077-1084 */
        }
        size_t n = 0;
        for (int o = 0; o < out_n; o++) n += last[o] >= first[o] ? (size_t)(last[o] - first[o] + 1) : 0;
        g->src = (int *)malloc(sizeof(int) * (n + 1));
        g->w = (float *)malloc(sizeof(float) * (n + 1));
        if (!g->src || !g->w) goto done;
        int e = 0;
        for (int o = 0; o < out_n; o++) {
            g->start[o] = e;
            for (int i = first[o]; i <= last[o]; i++) {
                const float wv = rows[(size_t)TAPS * o + (i - first[o])];
                if (wv == 0.0f) continue; /* adds +0 in the library */
                g->src[e] = clamp_index(i, in_n);
                g->w[e++] = wv;
            }
        }
        g->start[out_n] = e;
        g->n_entries = e;
    } else {
        const float reach = 2.0f / ratio; /* kernel support in input pixels, :1216 */
        for (int r = 0; r < nrows; r++) {
            const float ic = (float)(r - margin) + 0.5f;
            const float lo = (ic - reach) * ratio - shift, hi = (ic + reach) * ratio - shift; /* :1025-1030 */
            const float at = ic * ratio - shift;                                             /* :1032 */
            const int a = (int)floor(lo + 0.5), b = (int)floor(hi - 0.5);
            float *row = rows + (size_t)TAPS * r;
            for (int t = 0; t <= b - a; t++) row[t] = w_mitchell(((float)(t + a) + 0.5f) - at) * ratio; /* :1098-1103 */
            first[r] = a;
            last[r] = b;
            for (int t = b - a; t >= 0 && !row[t]; t--) last[r] = a + t - 1; /* :1107-1114 */
        }
        for (int o = 0; o < out_n; o++) { /* weights arriving at one output sum to 1, :1124-1150 */
            float sum = 0;
            for (int r = 0; r < nrows; r++) {
                if (o >= first[r] && o <= last[r]) sum += rows[(size_t)TAPS * r + (o - first[r])];
                else if (o < first[r]) break;
            }
            const float norm = 1 / sum;
            for (int r = 0; r < nrows; r++) {
                if (o >= first[r] && o <= last[r]) rows[(size_t)TAPS * r + (o - first[r])] *= norm;
                else if (o < first[r]) break;
            }
        }
        for (int r = 0; r < nrows; r++) { /* windows trimmed to the image, rows shifted, :1154-1187 */
            int skip = 0;
            while (rows[(size_t)TAPS * r + skip] == 0) skip++; /* an all-zero row scans on: harmless, it only ever adds +0 */
            first[r] += skip;
            while (first[r] < 0) {
                first[r]++;
                skip++;
            }
            const int span = last[r] - first[r] + 1, n = TAPS < span ? TAPS : span;
            for (int t = 0; t < n && t + skip < TAPS; t++) rows[(size_t)TAPS * r + t] = rows[(size_t)TAPS * r + t + skip];
        }
        for (int r = 0; r < nrows; r++)
            if (last[r] > out_n - 1) last[r] = out_n - 1; /* :1190-1191 */
        size_t n = 0;
        for (int r = 0; r < nrows; r++) n += last[r] >= first[r] ? (size_t)(last[r] - first[r] + 1) : 0;
        g->src = (int *)malloc(sizeof(int) * (n + 1));
        g->w = (float *)malloc(sizeof(float) * (n + 1));
        if (!g->src || !g->w) goto done;
        int e = 0;
        for (int o = 0; o < out_n; o++) { /* the library visits inputs in increasing order: so does the list */
            g->start[o] = e;
            for (int r = 0; r < nrows; r++) {
                if (o < first[r] || o > last[r]) continue;
                const float wv = rows[(size_t)TAPS * r + (o - first[r])];
                if (wv == 0.0f) continue;
                g->src[e] = clamp_index(r - margin, in_n);
                g->w[e++] = wv;
            }
        }
        g->start[out_n] = e;
        g->n_entries = e;
    }
    g->n_out = out_n;
    rc = 0;
done:
    free(rows);
    free(first);
    free(last);
    if (rc) gather_free(g);
    return rc;
}

/* ---- one cached geometry: tables on the device */
typedef struct {
    int w, h, tw, th, nw, nh, px, py;
    int max_cols, max_rows; /* largest source region of a 16 x 16 output tile (for the LDS-tiled kernel) */
    int n_xtaps, max_ytaps; /* entries of the horizontal gather list / the longest vertical list of one output row (the strip kernel stages them in LDS) */
    void *dev; /* [xstart][xsrc][xw][ystart][ysrc][yw] */
    size_t off[6];
} geom_t;
static geom_t g_geom;

static int geometry(int w, int h, int tw, int th, geom_t **out) {
    if (g_geom.dev && g_geom.w == w && g_geom.h == h && g_geom.tw == tw && g_geom.th == th) {
        *out = &g_geom;
        return 0;
    }
    const float scale = fminf((float)tw / w, (float)th / h); /* mars_yolo_test.c:47 */
    const int nw = (int)(w * scale), nh = (int)(h * scale);  /* :48 */
    if (nw <= 0 || nh <= 0 || nw > tw || nh > th) return -1;
    gather_t gx, gy;
    if (build_axis(w, nw, &gx)) return -1;
    if (build_axis(h, nh, &gy)) { gather_free(&gx); return -1; }
    /* tiles start at multiples of 16 of the TARGET, i.e. at (16 i - pad) of the resized image */
    const int px0 = (tw - nw) / 2, py0 = (th - nh) / 2;
    int max_cols = 1, max_rows = 1;
    for (int t0 = -(px0 % 16); t0 < nw; t0 += 16) {
        const int a = t0 < 0 ? 0 : t0, b = t0 + 16 < nw ? t0 + 16 : nw;
        if (b <= a) continue;
        int lo = gx.src[gx.start[a]], hi = lo;
        for (int o = a; o < b; o++) {
            if (gx.src[gx.start[o]] < lo) lo = gx.src[gx.start[o]];
            if (gx.src[gx.start[o + 1] - 1] > hi) hi = gx.src[gx.start[o + 1] - 1];
        }
        const int span = hi - lo + 1;
        if (span > max_cols) max_cols = span;
    }
    for (int t0 = -(py0 % 16); t0 < nh; t0 += 16) {
        const int a = t0 < 0 ? 0 : t0, b = t0 + 16 < nh ? t0 + 16 : nh;
        if (b <= a) continue;
        int lo = gy.src[gy.start[a]], hi = lo;
        for (int o = a; o < b; o++) {
            if (gy.src[gy.start[o]] < lo) lo = gy.src[gy.start[o]];
            if (gy.src[gy.start[o + 1] - 1] > hi) hi = gy.src[gy.start[o + 1] - 1];
        }
        const int span = hi - lo + 1;
        if (span > max_rows) max_rows = span;
    }
    const size_t sz[6] = {sizeof(int) * ((size_t)nw + 1), sizeof(int) * (size_t)(gx.n_entries + 1), sizeof(float) * (size_t)(gx.n_entries + 1),
                          sizeof(int) * ((size_t)nh + 1), sizeof(int) * (size_t)(gy.n_entries + 1), sizeof(float) * (size_t)(gy.n_entries + 1)};
    const void *srcs[6] = {gx.start, gx.src, gx.w, gy.start, gy.src, gy.w};
    const int n_xtaps = gx.n_entries;
    int max_ytaps = 0;
    for (int o = 0; o < nh; o++)
        if (gy.start[o + 1] - gy.start[o] > max_ytaps) max_ytaps = gy.start[o + 1] - gy.start[o];
    size_t total = 0, off[6];
    for (int i = 0; i < 6; i++) {
        off[i] = total;
        total += (sz[i] + 255) & ~(size_t)255;
    }
    int rc = -1;
    void *dev = mhip_malloc(total);
    if (dev) {
        rc = 0;
        for (int i = 0; i < 6 && !rc; i++) rc = mhip_h2d_async((char *)dev + off[i], srcs[i], sz[i]);
        if (!rc) rc = mhip_sync();
    }
    gather_free(&gx);
    gather_free(&gy);
    if (rc) {
        if (dev) mhip_free(dev);
        return -1;
    }
    if (g_geom.dev) mhip_free(g_geom.dev);
    g_geom.w = w; g_geom.h = h; g_geom.tw = tw; g_geom.th = th;
    g_geom.nw = nw; g_geom.nh = nh;
    g_geom.px = (tw - nw) / 2; g_geom.py = (th - nh) / 2; /* :49 */
    g_geom.max_cols = max_cols; g_geom.max_rows = max_rows;
    g_geom.n_xtaps = n_xtaps; g_geom.max_ytaps = max_ytaps;
    g_geom.dev = dev;
    memcpy(g_geom.off, off, sizeof(off));
    *out = &g_geom;
    return 0;
}

static int run_letterbox(const geom_t *g, const uint8_t *rgb_dev, size_t rgb_stride, int8_t *out_dev, size_t out_stride,
                         int frames, int nhwc) {
    mhip_letterbox_t p;
    memset(&p, 0, sizeof(p));
    p.rgb = rgb_dev; p.rgb_stride = rgb_stride;
    p.out = out_dev; p.out_stride = out_stride;
    p.frames = frames; p.w = g->w; p.h = g->h; p.tw = g->tw; p.th = g->th; p.nhwc = nhwc;
    p.nw = g->nw; p.nh = g->nh; p.px = g->px; p.py = g->py;
    p.max_cols = g->max_cols; p.max_rows = g->max_rows;
    p.n_xtaps = g->n_xtaps; p.max_ytaps = g->max_ytaps;
    { /* tests: MARS_HIP_LETTERBOX_FORM = 1 / 2 forces the 16 x 16-tile / the one-thread-per-pixel kernel (all three write the same bytes) */
        const char *e = getenv("MARS_HIP_LETTERBOX_FORM");
        p.form = e ? atoi(e) : 0;
    }
    const char *b = (const char *)g->dev;
    p.xstart = (const int *)(b + g->off[0]); p.xsrc = (const int *)(b + g->off[1]); p.xw = (const float *)(b + g->off[2]);
    p.ystart = (const int *)(b + g->off[3]); p.ysrc = (const int *)(b + g->off[4]); p.yw = (const float *)(b + g->off[5]);
    return mhip_letterbox(&p);
}

/* host pointers in and out, one frame: the reference's load_image() on an already decoded image */
int mars_yolo_letterbox(const unsigned char *rgb, int w, int h, int tw, int th, int nhwc, signed char *out) {
    if (!rgb || !out || w <= 0 || h <= 0 || tw <= 0 || th <= 0 || !mhip_ready()) return -1;
    geom_t *g;
    if (geometry(w, h, tw, th, &g)) return -1;
    const size_t in_b = (size_t)w * h * 3, out_b = (size_t)tw * th * 3;
    uint8_t *d = (uint8_t *)mhip_malloc(((in_b + 255) & ~(size_t)255) + out_b);
    if (!d) return -1;
    int8_t *dout = (int8_t *)(d + ((in_b + 255) & ~(size_t)255));
    int rc = mhip_h2d_async(d, rgb, in_b);
    if (!rc) rc = run_letterbox(g, d, in_b, dout, out_b, 1, nhwc);
    if (!rc) rc = mhip_d2h_async(out, dout, out_b);
    if (mhip_sync()) rc = -1;
    mhip_free(d);
    return rc ? -1 : 0;
}

/* build (and cache) the gather tables of a geometry ahead of the first frame: the pipelined camera path calls it when the pipe opens */
int mars_preproc_prepare(int w, int h, int tw, int th) {
    geom_t *g;
    return geometry(w, h, tw, th, &g);
}

/* geometry + target checks shared by the host and the device form */
static mars_error_t preproc_target(mars_model_t *model, int input_index, int w, int h, int first_frame, int frames, geom_t **g,
                                   int8_t **dst, size_t *dst_stride, int *nhwc_out) {
    if (!model || w <= 0 || h <= 0 || frames <= 0 || first_frame < 0) return MARS_ERR_INVALID_FILE;
    mars_model_ext_t *m = (mars_model_ext_t *)model;
    if (!m->act_dev || !mhip_ready()) return MARS_ERR_NNA_INIT_FAILED;
    if (input_index < 0 || (uint32_t)input_index >= model->header.num_inputs || first_frame + frames > m->batch)
        return MARS_ERR_INVALID_TENSOR;
    const uint32_t tid = model->header.input_tensor_ids[input_index]; /* an index, as in mars_get_input() */
    if (tid >= model->header.num_tensors) return MARS_ERR_INVALID_TENSOR;
    const int ti = (int)tid;
    const mars_tensor_t *t = &model->tensors[ti].desc;
    const int nhwc = t->format == MARS_FORMAT_NHWC;
    const int th = nhwc ? t->shape[1] : t->shape[2], tw = nhwc ? t->shape[2] : t->shape[3];
    const int ch = nhwc ? t->shape[3] : t->shape[1];
    if (ch != 3 || t->dtype != MARS_DTYPE_INT8 || tw <= 0 || th <= 0 || !m->mt[ti].dev ||
        m->mt[ti].stride < (size_t)tw * th * 3)
        return MARS_ERR_INVALID_TENSOR;
    if (geometry(w, h, tw, th, g)) return MARS_ERR_LAYER_FAILED;
    *dst = (int8_t *)m->mt[ti].dev + (size_t)first_frame * m->mt[ti].stride;
    *dst_stride = m->mt[ti].stride;
    *nhwc_out = nhwc;
    return MARS_OK;
}

/* camera batch: `frames` RGB frames of w x h (host, contiguous) -> frames [first_frame, first_frame + frames) of
 * graph input `input_index` on the device, in the layout the input tensor's format tag asks for
 * (mars_yolo_test.c:157-165).  The frames are staged through one device buffer; no host-side pixel work. */
mars_error_t mars_hip_preprocess(mars_model_t *model, int input_index, const unsigned char *rgb_frames, int w, int h,
                                 int first_frame, int frames) {
    if (!rgb_frames) return MARS_ERR_INVALID_FILE;
    geom_t *g;
    int8_t *dst;
    size_t dst_stride;
    int nhwc;
    const mars_error_t e = preproc_target(model, input_index, w, h, first_frame, frames, &g, &dst, &dst_stride, &nhwc);
    if (e != MARS_OK) return e;
    const size_t in_b = (size_t)w * h * 3;
    uint8_t *d = (uint8_t *)mhip_malloc(in_b * (size_t)frames);
    if (!d) return MARS_ERR_ALLOC_FAILED;
    int rc = mhip_h2d_async(d, rgb_frames, in_b * (size_t)frames);
    if (!rc) rc = run_letterbox(g, d, in_b, dst, dst_stride, frames, nhwc);
    if (mhip_sync()) rc = -1;
    mhip_free(d);
    return rc ? MARS_ERR_LAYER_FAILED : MARS_OK;
}

/* the same with the frames already in device memory: enqueued on the current stream, not synchronised (the pipelined camera
 * path of mars_pipe.c, or a caller whose capture hardware writes into HBM) */
mars_error_t mars_hip_preprocess_device(mars_model_t *model, int input_index, const void *rgb_dev, int w, int h,
                                        int first_frame, int frames) {
    if (!rgb_dev) return MARS_ERR_INVALID_FILE;
    geom_t *g;
    int8_t *dst;
    size_t dst_stride;
    int nhwc;
    const mars_error_t e = preproc_target(model, input_index, w, h, first_frame, frames, &g, &dst, &dst_stride, &nhwc);
    if (e != MARS_OK) return e;
    return run_letterbox(g, (const uint8_t *)rgb_dev, (size_t)w * h * 3, dst, dst_stride, frames, nhwc) ? MARS_ERR_LAYER_FAILED : MARS_OK;
}
