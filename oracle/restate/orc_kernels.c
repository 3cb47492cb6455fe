/*
 * orc_kernels.c -- CPU restatement of the reference's layer arithmetic.
 * TEST INFRASTRUCTURE (see orc.h).  Written from the behaviour of the
 * reference sources cited per function; compiled with -ffp-contract=off so
 * that no multiply-add is fused (the reference's results on x86-64 depend on
 * separate roundings).
 */
#include <math.h>
#include <string.h>

#include "orc.h"

int32_t orc_trunc_x86(float x) {
    /* representable window of cvttss2si: [-2^31, 2^31) */
    if (x >= -2147483648.0f && x < 2147483648.0f) return (int32_t)x;
    return INT32_MIN; /* "integer indefinite": too large, too small or NaN */
}

static inline int8_t sat8(int32_t v) {
    if (v > 127) return 127;
    if (v < -128) return -128;
    return (int8_t)v;
}

/* half-away-from-zero requantisation of a conv accumulator
 * (reference mxu_conv.c:663-665 / :750-752) */
static inline int8_t requant_conv(int32_t acc, float cs) {
    float scaled = (float)acc * cs;
    float biased = scaled + (scaled >= 0 ? 0.5f : -0.5f);
    return sat8(orc_trunc_x86(biased));
}

/* "+0.5 then truncate" requantisation used by the element-wise layers
 * (reference mars_runtime.c:764, :831, :898, :1147) */
static inline int8_t requant_half_up(float v) { return sat8(orc_trunc_x86(v + 0.5f)); }

/* ------------------------------------------------------------------ conv */

void orc_conv_i8_nhwc(const int8_t *in, const int8_t *w, const int32_t *bias, int8_t *out,
                      const orc_conv_geom_t *g, float in_scale, float w_scale, float out_scale) {
    /* reference mxu_conv.c:713-757.  K order (kh, kw, ic); taps outside the
     * image contribute nothing; int32 accumulation so order is immaterial. */
    const float cs = (in_scale * w_scale) / out_scale;
    const int taps_c = g->kh * g->kw * g->in_c;
    for (int oy = 0; oy < g->out_h; oy++) {
        for (int ox = 0; ox < g->out_w; ox++) {
            int8_t *dst = out + ((size_t)oy * g->out_w + ox) * g->out_c;
            const int y0 = oy * g->stride_h - g->pad_top;
            const int x0 = ox * g->stride_w - g->pad_left;
            for (int oc = 0; oc < g->out_c; oc++) {
                const int8_t *wk = w + (size_t)oc * taps_c;
                int32_t acc = bias ? bias[oc] : 0;
                for (int ky = 0; ky < g->kh; ky++) {
                    const int iy = y0 + ky;
                    if (iy < 0 || iy >= g->in_h) continue;
                    for (int kx = 0; kx < g->kw; kx++) {
                        const int ix = x0 + kx;
                        if (ix < 0 || ix >= g->in_w) continue;
                        const int8_t *px = in + ((size_t)iy * g->in_w + ix) * g->in_c;
                        const int8_t *wt = wk + ((size_t)ky * g->kw + kx) * g->in_c;
                        int32_t s = 0;
                        for (int ic = 0; ic < g->in_c; ic++) s += (int32_t)px[ic] * (int32_t)wt[ic];
                        acc += s;
                    }
                }
                dst[oc] = requant_conv(acc, cs);
            }
        }
    }
}

void orc_conv_i8_nchw(const int8_t *in, const int8_t *w, const int32_t *bias, int8_t *out,
                      const orc_conv_geom_t *g, float in_scale, float w_scale, float out_scale) {
    /* reference mxu_conv.c:630-670.  K order (ic, kh, kw). */
    const float cs = (in_scale * w_scale) / out_scale;
    const size_t plane_in = (size_t)g->in_h * g->in_w;
    const size_t plane_out = (size_t)g->out_h * g->out_w;
    const int per_ic = g->kh * g->kw;
    for (int oc = 0; oc < g->out_c; oc++) {
        const int8_t *wk = w + (size_t)oc * g->in_c * per_ic;
        for (int oy = 0; oy < g->out_h; oy++) {
            const int y0 = oy * g->stride_h - g->pad_top;
            for (int ox = 0; ox < g->out_w; ox++) {
                const int x0 = ox * g->stride_w - g->pad_left;
                int32_t acc = bias ? bias[oc] : 0;
                for (int ky = 0; ky < g->kh; ky++) {
                    const int iy = y0 + ky;
                    if (iy < 0 || iy >= g->in_h) continue;
                    for (int kx = 0; kx < g->kw; kx++) {
                        const int ix = x0 + kx;
                        if (ix < 0 || ix >= g->in_w) continue;
                        const int8_t *px = in + (size_t)iy * g->in_w + ix;
                        const int8_t *wt = wk + ky * g->kw + kx;
                        for (int ic = 0; ic < g->in_c; ic++)
                            acc += (int32_t)px[ic * plane_in] * (int32_t)wt[ic * per_ic];
                    }
                }
                out[oc * plane_out + (size_t)oy * g->out_w + ox] = requant_conv(acc, cs);
            }
        }
    }
}

void orc_conv_f32_nchw(const float *in, const float *w, const float *bias, float *out,
                       const orc_conv_geom_t *g) {
    /* reference mxu_conv.c:673-710.  The summation ORDER is part of the
     * contract: start from the bias, then (ic, kh, kw) ascending, one rounded
     * multiply and one rounded add per in-image tap. */
    const size_t plane_in = (size_t)g->in_h * g->in_w;
    const size_t plane_out = (size_t)g->out_h * g->out_w;
    const int per_ic = g->kh * g->kw;
    for (int oc = 0; oc < g->out_c; oc++) {
        const float *wk = w + (size_t)oc * g->in_c * per_ic;
        for (int oy = 0; oy < g->out_h; oy++) {
            for (int ox = 0; ox < g->out_w; ox++) {
                float acc = bias ? bias[oc] : 0.0f;
                for (int ic = 0; ic < g->in_c; ic++) {
                    const float *plane = in + ic * plane_in;
                    for (int ky = 0; ky < g->kh; ky++) {
                        const int iy = oy * g->stride_h - g->pad_top + ky;
                        for (int kx = 0; kx < g->kw; kx++) {
                            const int ix = ox * g->stride_w - g->pad_left + kx;
                            if (iy < 0 || iy >= g->in_h || ix < 0 || ix >= g->in_w) continue;
                            float prod = plane[(size_t)iy * g->in_w + ix] * wk[(ic * g->kh + ky) * g->kw + kx];
                            acc = acc + prod;
                        }
                    }
                }
                out[oc * plane_out + (size_t)oy * g->out_w + ox] = acc;
            }
        }
    }
}

/* ------------------------------------------------------------ elementwise */

void orc_relu_bytes(int8_t *buf, size_t n) {
    /* reference mars_runtime.c:700-707: byte-wise, whatever the dtype */
    for (size_t i = 0; i < n; i++)
        if (buf[i] < 0) buf[i] = 0;
}

void orc_sigmoid_i8(const int8_t *in, int8_t *out, size_t n, float in_scale, float out_scale) {
    const float os = out_scale > 0 ? out_scale : 1.0f;
    for (size_t i = 0; i < n; i++) {
        float x = (float)in[i] * in_scale;
        float y = 1.0f / (1.0f + expf(-x));
        out[i] = requant_half_up(y / os);
    }
}

void orc_sigmoid_f32(const float *in, float *out, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = 1.0f / (1.0f + expf(-in[i]));
}

void orc_binary_i8(int is_mul, const int8_t *a, const int8_t *b, int8_t *out, size_t n, float sa,
                   float sb, float so) {
    const float inv = 1.0f / (so > 0 ? so : 1.0f);
    for (size_t i = 0; i < n; i++) {
        float va = (float)a[i] * sa;
        float vb = (float)b[i] * sb;
        float y = is_mul ? va * vb : va + vb;
        out[i] = requant_half_up(y * inv);
    }
}

void orc_binary_f32(int is_mul, const float *a, const float *b, float *out, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = is_mul ? a[i] * b[i] : a[i] + b[i];
}

void orc_relu_i8(const int8_t *in, int8_t *out, size_t n, int leaky) {
    for (size_t i = 0; i < n; i++) {
        int8_t v = in[i];
        if (v > 0) {
            out[i] = v;
        } else if (leaky) {
            int32_t t = orc_trunc_x86((float)v * 0.01f); /* slope is fixed, :1064 */
            out[i] = (int8_t)(t < -128 ? -128 : t);
        } else {
            out[i] = 0;
        }
    }
}

void orc_relu_f32(const float *in, float *out, size_t n, int leaky) {
    const float alpha = leaky ? 0.01f : 0.0f;
    for (size_t i = 0; i < n; i++) out[i] = in[i] > 0.0f ? in[i] : in[i] * alpha;
}

/* --------------------------------------------------------- data movement */

void orc_maxpool_i8(const int8_t *in, int8_t *out, int in_h, int in_w, int ch, int out_h,
                    int out_w, int kh, int kw, int sh, int sw) {
    /* reference :919-957: no padding, window clipped at bottom/right only,
     * identity element -128 (an empty window therefore yields -128) */
    for (int oy = 0; oy < out_h; oy++)
        for (int ox = 0; ox < out_w; ox++)
            for (int c = 0; c < ch; c++) {
                int8_t best = -128;
                for (int ky = 0; ky < kh; ky++) {
                    int iy = oy * sh + ky;
                    if (iy >= in_h) break;
                    for (int kx = 0; kx < kw; kx++) {
                        int ix = ox * sw + kx;
                        if (ix >= in_w) break;
                        int8_t v = in[((size_t)iy * in_w + ix) * ch + c];
                        if (v > best) best = v;
                    }
                }
                out[((size_t)oy * out_w + ox) * ch + c] = best;
            }
}

void orc_concat_slice_i8(const int8_t *in, int8_t *out, int out_h, int out_w, int in_c, int out_c,
                         int ch_off) {
    /* one input of reference :978-997; the source is walked with the OUTPUT's
     * spatial extent and its own channel count */
    for (int y = 0; y < out_h; y++)
        for (int x = 0; x < out_w; x++) {
            const int8_t *s = in + ((size_t)y * out_w + x) * in_c;
            int8_t *d = out + ((size_t)y * out_w + x) * out_c + ch_off;
            for (int c = 0; c < in_c; c++) d[c] = s[c];
        }
}

void orc_upsample_i8(const int8_t *in, int8_t *out, int in_h, int in_w, int ch, int out_h,
                     int out_w, int scale_h, int scale_w) {
    for (int oy = 0; oy < out_h; oy++) {
        int iy = oy / scale_h;
        if (iy >= in_h) iy = in_h - 1;
        for (int ox = 0; ox < out_w; ox++) {
            int ix = ox / scale_w;
            if (ix >= in_w) ix = in_w - 1;
            memcpy(out + ((size_t)oy * out_w + ox) * ch, in + ((size_t)iy * in_w + ix) * ch, (size_t)ch);
        }
    }
}

void orc_batchnorm_i8(const int8_t *in, int8_t *out, int n, int c, int h, int w, const float *s,
                      const float *b, float in_scale, float out_scale) {
    const float is = in_scale > 0 ? in_scale : 1.0f;
    const float os = out_scale > 0 ? out_scale : 1.0f;
    const size_t hw = (size_t)h * w;
    for (int ni = 0; ni < n; ni++)
        for (int ci = 0; ci < c; ci++) {
            const float sc = s ? s[ci] : 1.0f;
            const float bi = b ? b[ci] : 0.0f;
            const size_t base = ((size_t)ni * c + ci) * hw;
            for (size_t i = 0; i < hw; i++) {
                float x = (float)in[base + i] * is;
                float m = x * sc;
                float y = m + bi;
                out[base + i] = requant_half_up(y / os);
            }
        }
}

void orc_batchnorm_f32(const float *in, float *out, int n, int c, int h, int w, const float *s,
                       const float *b) {
    const size_t hw = (size_t)h * w;
    for (int ni = 0; ni < n; ni++)
        for (int ci = 0; ci < c; ci++) {
            const float sc = s ? s[ci] : 1.0f;
            const float bi = b ? b[ci] : 0.0f;
            const size_t base = ((size_t)ni * c + ci) * hw;
            for (size_t i = 0; i < hw; i++) {
                float m = in[base + i] * sc;
                out[base + i] = m + bi;
            }
        }
}
