/*
 * orc_resize.c -- CPU restatement of the reference's image front-end
 * (reference src/mars/mars_yolo_test.c:40-77, load_image(), minus the file decode).
 *
 * TEST INFRASTRUCTURE (see orc.h).
 *
 * load_image() letterbox-resizes with stbir_resize_uint8() of the stb_image_resize header the reference
 * vendors (include/stb/stb_image_resize.h, v0.9x, third party, public domain): default filters
 * (Catmull-Rom when an axis grows, Mitchell-Netravali when it shrinks or stays), clamped edges, linear
 * colour space, float arithmetic.  What is restated here is that library's published algorithm, operation
 * for operation, because the result is compared bit for bit:
 *   - per axis a table of float coefficients: for a growing axis each OUTPUT pixel gathers the input pixels
 *     inside its kernel support and the coefficients are scaled to sum 1 (stb_image_resize.h:1008-1085,
 *     1194-1213); for a shrinking axis each INPUT pixel (margins included) scatters to the output pixels it
 *     touches, coefficients = kernel * scale, then every output's incoming coefficients are scaled to sum 1
 *     (:1023-1035, 1087-1192, 1214-1230);
 *   - pixels: value/255 -> horizontal pass -> vertical pass -> clamp to [0,1] -> *255 -> +0.5 (in double)
 *     -> truncate (:1243-1283, 1441-1652, 1866-2061, 1726-1740).  Both passes accumulate `acc += v * c`
 *     in increasing source order starting from 0, whether the library gathers or scatters, so one gather
 *     formulation with the entries in increasing source order reproduces either.  Terms with a zero
 *     coefficient add +0 and are dropped.
 * Checked against the reference compiled in place (oracle/_ref, ref_o3_load_image) over many geometries:
 * tests/test_oracle.py::test_letterbox_*.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orc.h"

static float kern_catmullrom(float x) { /* stb_image_resize.h:810-822 */
    x = (float)fabs(x);
    if (x < 1.0f) return 1 - x * x * (2.5f - 1.5f * x);
    else if (x < 2.0f) return 2 - x * (4 + x * (0.5f * x - 2.5f));
    return 0.0f;
}
static float kern_mitchell(float x) { /* :824-836 */
    x = (float)fabs(x);
    if (x < 1.0f) return (16 + x * x * (21 * x - 36)) / 18;
    else if (x < 2.0f) return (32 + x * (-60 + x * (36 - 7 * x))) / 18;
    return 0.0f;
}

/* gather list of one axis: output i sums src[k]*coef[k] for k in [start[i], start[i+1]) , sources increasing */
typedef struct {
    int n_out;
    int *start;
    int *src;
    float *coef;
} axis_t;

static void axis_free(axis_t *a) {
    free(a->start);
    free(a->src);
    free(a->coef);
    memset(a, 0, sizeof(*a));
}

static int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

#define WIDTH 4 /* coefficients per contributor for both default filters (support 2): :900-906 */

static int axis_build(axis_t *a, int in_size, int out_size) {
    const float scale = ((float)out_size / in_size) / (1.0f - 0.0f); /* :2221-2222, s0 = 0, s1 = 1 */
    const float shift = 0.0f * out_size / (1.0f - 0.0f);            /* :2224 */
    const int up = scale > 1;                                       /* :864-867 */
    memset(a, 0, sizeof(*a));
    a->n_out = out_size;
    a->start = (int *)calloc((size_t)out_size + 1, sizeof(int));
    if (!a->start) return -1;
    if (up) {
        /* every output pixel: which input pixels lie under its kernel, Catmull-Rom weights, normalised */
        a->src = (int *)malloc(sizeof(int) * (size_t)out_size * (WIDTH + 2));
        a->coef = (float *)malloc(sizeof(float) * (size_t)out_size * (WIDTH + 2));
        float *flat = (float *)calloc((size_t)out_size * WIDTH + 16, sizeof(float)); /* the library's flat table */
        int *n0s = (int *)malloc(sizeof(int) * (size_t)out_size), *n1s = (int *)malloc(sizeof(int) * (size_t)out_size);
        if (!a->src || !a->coef || !flat || !n0s || !n1s) return -1;
        const float radius = 2.0f * scale; /* support(1/scale) * scale, :1200 */
        for (int n = 0; n < out_size; n++) {
            const float c = (float)n + 0.5f;
            const float lb = c - radius, ub = c + radius;
            const float in_lb = (lb + shift) / scale, in_ub = (ub + shift) / scale;
            const float center = (c + shift) / scale;
            int first = (int)floor(in_lb + 0.5), last = (int)floor(in_ub - 0.5); /* double arithmetic, :1018-1019 */
            float *g = flat + (size_t)WIDTH * n;
            float total = 0;
            int n0 = first, n1 = last;
            for (int i = 0; i <= last - first; i++) {
                const float pc = (float)(i + first) + 0.5f;
                g[i] = kern_catmullrom(center - pc);
                if (i == 0 && !g[i]) { /* leading zero: move the window (:1056-1061) */
                    n0 = ++first;
                    i--;
                    continue;
                }
                total += g[i];
            }
            const float fs = 1 / total;
            for (int i = 0; i <= last - first; i++) g[i] *= fs;
            for (int i = last - first; i >= 0; i--) {
                if (g[i]) break;
                n1 = n0 + i - 1;
            }
            n0s[n] = n0;
            n1s[n] = n1;
        }
        int w = 0;
        for (int n = 0; n < out_size; n++) {
            a->start[n] = w;
            for (int k = n0s[n]; k <= n1s[n]; k++) {
                const float cf = flat[(size_t)WIDTH * n + (k - n0s[n])]; /* read AFTER all groups were written */
                if (cf == 0.0f) continue;
                a->src[w] = clampi(k, in_size - 1);
                a->coef[w++] = cf;
            }
        }
        a->start[out_size] = w;
        free(flat);
        free(n0s);
        free(n1s);
        return 0;
    }
    /* shrinking (or equal) axis: every input pixel, margins included, scatters Mitchell weights */
    const float radius_in = 2.0f / scale;                            /* :1216 */
    const int pixel_width = (int)ceil(2.0f * 2 / scale);            /* :889 */
    const int margin = pixel_width / 2;
    const int ncon = in_size + margin * 2;
    float *flat = (float *)calloc((size_t)ncon * WIDTH + 64, sizeof(float));
    int *n0s = (int *)malloc(sizeof(int) * (size_t)ncon), *n1s = (int *)malloc(sizeof(int) * (size_t)ncon);
    if (!flat || !n0s || !n1s) return -1;
    for (int i = 0; i < 64; i++) flat[(size_t)ncon * WIDTH + i] = 1.0f; /* what lies behind the table never matters: see below */
    for (int n = 0; n < ncon; n++) {
        const int nadj = n - margin;
        const float c = (float)nadj + 0.5f;
        const float lb = c - radius_in, ub = c + radius_in;
        const float out_lb = lb * scale - shift, out_ub = ub * scale - shift;
        const float center = c * scale - shift;
        const int first = (int)floor(out_lb + 0.5), last = (int)floor(out_ub - 0.5);
        float *g = flat + (size_t)WIDTH * n;
        int n1 = last;
        for (int i = 0; i <= last - first; i++) {
            const float pc = (float)(i + first) + 0.5f;
            const float x = pc - center;
            g[i] = kern_mitchell(x) * scale;
        }
        for (int i = last - first; i >= 0; i--) {
            if (g[i]) break;
            n1 = first + i - 1;
        }
        n0s[n] = first;
        n1s[n] = n1;
    }
    /* every output's incoming coefficients scaled to sum 1 (:1117-1150) */
    for (int i = 0; i < out_size; i++) {
        float total = 0;
        for (int j = 0; j < ncon; j++) {
            if (i >= n0s[j] && i <= n1s[j]) total += flat[(size_t)WIDTH * j + (i - n0s[j])];
            else if (i < n0s[j]) break;
        }
        const float sc = 1 / total;
        for (int j = 0; j < ncon; j++) {
            if (i >= n0s[j] && i <= n1s[j]) flat[(size_t)WIDTH * j + (i - n0s[j])] *= sc;
            else if (i < n0s[j]) break;
        }
    }
    /* leading zeros and outputs left of the image dropped, table rows shifted (:1154-1187).  A row of zeros makes
     * the library scan into the following rows (and, for the last rows, past the table): the window start it
     * derives from that is irrelevant because such a row only ever adds +0, so the scan is simply stopped. */
    for (int j = 0; j < ncon; j++) {
        int skip = 0;
        while (flat[(size_t)WIDTH * j + skip] == 0) skip++;
        n0s[j] += skip;
        while (n0s[j] < 0) {
            n0s[j]++;
            skip++;
        }
        const int range = n1s[j] - n0s[j] + 1;
        const int max = WIDTH < range ? WIDTH : range;
        for (int i = 0; i < max; i++) {
            if (i + skip >= WIDTH) break;
            flat[(size_t)WIDTH * j + i] = flat[(size_t)WIDTH * j + i + skip];
        }
    }
    for (int j = 0; j < ncon; j++)
        if (n1s[j] > out_size - 1) n1s[j] = out_size - 1;
    /* invert to a gather list, sources in increasing order */
    size_t total_entries = 0;
    for (int j = 0; j < ncon; j++)
        if (n1s[j] >= n0s[j]) total_entries += (size_t)(n1s[j] - n0s[j] + 1);
    a->src = (int *)malloc(sizeof(int) * (total_entries + 1));
    a->coef = (float *)malloc(sizeof(float) * (total_entries + 1));
    if (!a->src || !a->coef) return -1;
    int w = 0;
    for (int k = 0; k < out_size; k++) {
        a->start[k] = w;
        for (int j = 0; j < ncon; j++) {
            if (k < n0s[j] || k > n1s[j]) continue;
            const int idx = k - n0s[j]; /* may run past this row into the next one, exactly as the library indexes */
            const float cf = flat[(size_t)WIDTH * j + idx];
            if (cf == 0.0f) continue;
            a->src[w] = clampi(j - margin, in_size - 1);
            a->coef[w++] = cf;
        }
    }
    a->start[out_size] = w;
    free(flat);
    free(n0s);
    free(n1s);
    return 0;
}

/* stbir_resize_uint8(in, w, h, 0, out, ow, oh, 0, 3): default filters, clamp, linear */
int orc_resize_rgb8(const uint8_t *in, int w, int h, uint8_t *out, int ow, int oh) {
    if (!in || !out || w <= 0 || h <= 0 || ow <= 0 || oh <= 0) return -1;
    axis_t ax, ay;
    if (axis_build(&ax, w, ow) || axis_build(&ay, h, oh)) return -1;
    float dec[256];
    for (int v = 0; v < 256; v++) dec[v] = ((float)v) / 255.0f; /* :1277 */
    float *hrow = (float *)malloc(sizeof(float) * (size_t)ow * 3);
    if (!hrow) return -1;
    for (int y = 0; y < oh; y++) {
        float *acc = (float *)calloc((size_t)ow * 3, sizeof(float));
        if (!acc) return -1;
        for (int e = ay.start[y]; e < ay.start[y + 1]; e++) {
            const uint8_t *row = in + (size_t)ay.src[e] * w * 3;
            for (int x = 0; x < ow; x++)
                for (int c = 0; c < 3; c++) {
                    float hsum = 0;
                    for (int k = ax.start[x]; k < ax.start[x + 1]; k++) hsum += dec[row[ax.src[k] * 3 + c]] * ax.coef[k];
                    hrow[x * 3 + c] = hsum;
                }
            const float vc = ay.coef[e];
            for (int i = 0; i < ow * 3; i++) acc[i] += hrow[i] * vc;
        }
        for (int i = 0; i < ow * 3; i++) {
            float v = acc[i];
            if (v < 0) v = 0;
            if (v > 1) v = 1;
            out[(size_t)y * ow * 3 + i] = (unsigned char)(int)((v * 255.0f) + 0.5); /* :1726,1735 */
        }
        free(acc);
    }
    free(hrow);
    axis_free(&ax);
    axis_free(&ay);
    return 0;
}

/* load_image() without the decode: RGB uint8 [h][w][3] -> int8 [th][tw][3] (nhwc) or [3][th][tw] */
int orc_letterbox(const uint8_t *rgb, int w, int h, int tw, int th, int nhwc, int8_t *out) {
    if (!rgb || !out || w <= 0 || h <= 0 || tw <= 0 || th <= 0) return -1;
    const float scale = fminf((float)tw / w, (float)th / h);       /* :47 */
    const int nw = (int)(w * scale), nh = (int)(h * scale);        /* :48 */
    const int px = (tw - nw) / 2, py = (th - nh) / 2;              /* :49 */
    if (nw <= 0 || nh <= 0) return -1;
    uint8_t *rsz = (uint8_t *)malloc((size_t)nw * nh * 3);
    if (!rsz) return -1;
    if (orc_resize_rgb8(rgb, w, h, rsz, nw, nh)) { free(rsz); return -1; }
    memset(out, -17, (size_t)tw * th * 3);                         /* :57 */
    for (int y = 0; y < nh; y++)
        for (int x = 0; x < nw; x++) {
            const int si = (y * nw + x) * 3, dy = y + py, dx = x + px;
            for (int c = 0; c < 3; c++) {
                const int8_t v = (int8_t)(rsz[si + c] - 128);
                if (nhwc) out[((size_t)dy * tw + dx) * 3 + c] = v;
                else out[(size_t)c * tw * th + (size_t)dy * tw + dx] = v;
            }
        }
    free(rsz);
    return 0;
}
