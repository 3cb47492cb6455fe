/*
 * mxu_compat.c -- the reference's directly callable kernels (include/mxu_ops.h)
 * with HOST pointers, served by the GPU: stage operands into HBM, launch the
 * same HIP kernels the graph executor uses, copy the result back.
 *
 * Replaces reference src/mars/mxu_ops.c:29-163 (mxu_init, f32 element-wise)
 * and the three externs of src/mars/mxu_conv.c (portable branch :628-758).
 * These are synchronous convenience entry points; they fail loudly (stderr,
 * output untouched) when no device is initialised -- there is no CPU path.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mars_internal.h"
#include "mxu_ops.h"
#include "nna.h"

static int g_mxu_flag = 0;

void mxu_init(void *nna_mem) {
    (void)nna_mem; /* no coprocessor control registers to program */
    g_mxu_flag = 1;
}
int mxu_is_initialized(void) { return g_mxu_flag; }

static int need_device(const char *who) {
    if (nna_is_ready()) return 0;
    if (nna_init() == NNA_SUCCESS) return 0;
    fprintf(stderr, "%s: no MI355X device available\n", who);
    return -1;
}

#define A256(x) (((size_t)(x) + 255) & ~(size_t)255)

static void f32_binary(int op, float *out, const float *a, const float *b, size_t count, const char *who) {
    if (count == 0 || need_device(who)) return;
    const size_t bytes = count * sizeof(float);
    uint8_t *d = (uint8_t *)mhip_malloc(A256(bytes) * 3);
    if (!d) { fprintf(stderr, "%s: device allocation failed\n", who); return; }
    float *da = (float *)d, *db = (float *)(d + A256(bytes)), *dc = (float *)(d + 2 * A256(bytes));
    int rc = mhip_h2d_async(da, a, bytes);
    if (!rc && b) rc = mhip_h2d_async(db, b, bytes);
    if (!rc) rc = op == 3 ? mhip_relu_f32(da, 0, dc, 0, 1, count, 0.0f) : mhip_binary_f32(op, da, 0, db, 0, dc, 0, 1, count);
    if (!rc) rc = mhip_d2h_async(out, dc, bytes);
    if (mhip_sync() || rc) fprintf(stderr, "%s: GPU execution failed: %s\n", who, mhip_last_error());
    mhip_free(d);
}

void mxu_mul_f32(float *out, const float *a, const float *b, size_t count) { f32_binary(1, out, a, b, count, "mxu_mul_f32"); }
void mxu_add_f32(float *out, const float *a, const float *b, size_t count) { f32_binary(0, out, a, b, count, "mxu_add_f32"); }
void mxu_sub_f32(float *out, const float *a, const float *b, size_t count) { f32_binary(2, out, a, b, count, "mxu_sub_f32"); }
void mxu_relu_f32(float *out, const float *in, size_t count) { f32_binary(3, out, in, NULL, count, "mxu_relu_f32"); }

static void conv_i8_host(int nchw, const signed char *input, int in_h, int in_w, int in_c, const signed char *weight,
                         int out_c, int kh, int kw, const int *bias, signed char *output, int out_h, int out_w,
                         int stride_h, int stride_w, int pad_top, int pad_left, float in_scale, float w_scale,
                         float out_scale, const char *who) {
    if (out_h <= 0 || out_w <= 0 || out_c <= 0) return;
    if (in_h <= 0 || in_w <= 0 || in_c <= 0 || kh <= 0 || kw <= 0 || !input || !weight || !output) {
        fprintf(stderr, "%s: invalid geometry\n", who);
        return;
    }
    if (need_device(who)) return;
    const int c_pad = nchw ? (in_c + 15) & ~15 : in_c;
    int row_pad, oc_pad, c_eff;
    mhip_conv_i8_pack_geom(c_pad, kw, out_c, &row_pad, &oc_pad, &c_eff);
    const size_t k64 = ((size_t)kh * row_pad + 63) & ~(size_t)63;
    const size_t in_b = (size_t)in_h * in_w * in_c, out_b = (size_t)out_h * out_w * out_c;
    const size_t w_b = (size_t)oc_pad * k64, scr_b = nchw ? (size_t)in_h * in_w * c_pad : 0;
    /* deep 3x3 stride-1 shapes: also the K-step image conv_i8_rows streams (launch variant 20; 0 bytes otherwise) */
    const size_t w3_b = nchw ? 0 : mhip_conv_i8_rows_pack(in_c, kh, kw, stride_h, stride_w, oc_pad, (int)k64, out_w, NULL, NULL);
    int8_t *hw3 = w3_b ? (int8_t *)malloc(w3_b) : NULL;
    int8_t *hw = (int8_t *)malloc(w_b);
    int32_t *hb = (int32_t *)calloc((size_t)oc_pad, 4);
    uint8_t *d = (uint8_t *)mhip_malloc(A256(in_b) + A256(out_b) + A256(w_b) + A256((size_t)oc_pad * 4) + A256(scr_b) + A256(w3_b) + 256);
    if (!hw || !hb || !d || (w3_b && !hw3)) { fprintf(stderr, "%s: allocation failed\n", who); goto done; }
    mars_pack_conv_i8((const int8_t *)weight, (size_t)out_c * in_c * kh * kw, nchw, out_c, in_c, kh, kw, c_eff, row_pad,
                      oc_pad, hw);
    for (int oc = 0; bias && oc < out_c; oc++) hb[mhip_conv_i8_oc_row(oc, oc_pad)] = bias[oc];
    {
        int8_t *din = (int8_t *)d, *dout = din + A256(in_b), *dw = dout + A256(out_b);
        int32_t *db = (int32_t *)(dw + A256(w_b));
        int8_t *dscr = (int8_t *)db + A256((size_t)oc_pad * 4);
        int8_t *dw3 = dscr + A256(scr_b);
        int rc = mhip_h2d_async(din, input, in_b);
        if (!rc) rc = mhip_h2d_async(dw, hw, w_b);
        if (!rc && w3_b) {
            mhip_conv_i8_rows_pack(in_c, kh, kw, stride_h, stride_w, oc_pad, (int)k64, out_w, hw, hw3);
            rc = mhip_h2d_async(dw3, hw3, w3_b);
        }
        if (!rc) rc = mhip_h2d_async(db, hb, (size_t)oc_pad * 4);
        mhip_conv_i8_t p;
        memset(&p, 0, sizeof(p));
        p.in = din; p.in_c = in_c;
        if (!rc && nchw) {
            rc = mhip_nchw_to_nhwc_pad(din, 0, dscr, 0, 1, in_c, in_h * in_w, c_pad);
            p.in = dscr; p.in_c = c_pad;
        }
        p.out = dout; p.w = dw; p.w_rows = w3_b ? dw3 : NULL; p.bias = bias ? db : NULL; p.frames = 1;
        p.in_h = in_h; p.in_w = in_w; p.out_h = out_h; p.out_w = out_w; p.out_c = out_c;
        p.kh = kh; p.kw = kw; p.stride_h = stride_h; p.stride_w = stride_w; p.pad_top = pad_top; p.pad_left = pad_left;
        p.row_pad = row_pad; p.oc_pad = oc_pad;
        p.cs = (in_scale * w_scale) / out_scale; /* mxu_conv.c:639 / :722 */
        p.out_nchw = nchw;
        p.safe = mhip_conv_i8_is_safe(p.cs);
        if (!rc) rc = mhip_conv_i8(&p);
        if (!rc) rc = mhip_d2h_async(output, dout, out_b);
        if (mhip_sync() || rc) fprintf(stderr, "%s: GPU execution failed: %s\n", who, mhip_last_error());
    }
done:
    free(hw3);
    free(hw);
    free(hb);
    if (d) mhip_free(d);
}

void conv2d_int8_mxu(const signed char *input, int in_h, int in_w, int in_c, const signed char *weight, int out_c,
                     int kh, int kw, const int *bias, signed char *output, int out_h, int out_w, int stride_h,
                     int stride_w, int pad_top, int pad_left, float in_scale, float w_scale, float out_scale) {
    conv_i8_host(1, input, in_h, in_w, in_c, weight, out_c, kh, kw, bias, output, out_h, out_w, stride_h, stride_w,
                 pad_top, pad_left, in_scale, w_scale, out_scale, "conv2d_int8_mxu");
}

void conv2d_int8_nhwc_mxu(const signed char *input, int in_h, int in_w, int in_c, const signed char *weight, int out_c,
                          int kh, int kw, const int *bias, signed char *output, int out_h, int out_w, int stride_h,
                          int stride_w, int pad_top, int pad_left, float in_scale, float w_scale, float out_scale) {
    conv_i8_host(0, input, in_h, in_w, in_c, weight, out_c, kh, kw, bias, output, out_h, out_w, stride_h, stride_w,
                 pad_top, pad_left, in_scale, w_scale, out_scale, "conv2d_int8_nhwc_mxu");
}

void conv2d_float32_mxu(const float *input, int in_h, int in_w, int in_c, const float *weight, int out_c, int kh,
                        int kw, const float *bias, float *output, int out_h, int out_w, int stride_h, int stride_w,
                        int pad_top, int pad_left, float *scratch) {
    (void)scratch;
    const char *who = "conv2d_float32_mxu";
    if (out_h <= 0 || out_w <= 0 || out_c <= 0) return;
    if (in_h <= 0 || in_w <= 0 || in_c <= 0 || kh <= 0 || kw <= 0 || !input || !weight || !output) {
        fprintf(stderr, "%s: invalid geometry\n", who);
        return;
    }
    if (need_device(who)) return;
    const size_t in_b = (size_t)in_h * in_w * in_c * 4, out_b = (size_t)out_h * out_w * out_c * 4;
    const size_t w_b = (size_t)out_c * in_c * kh * kw * 4, b_b = (size_t)out_c * 4;
    uint8_t *d = (uint8_t *)mhip_malloc(A256(in_b) + A256(out_b) + A256(w_b) + A256(b_b));
    if (!d) { fprintf(stderr, "%s: device allocation failed\n", who); return; }
    float *din = (float *)d, *dout = (float *)(d + A256(in_b)), *dw = (float *)(d + A256(in_b) + A256(out_b));
    float *db = (float *)(d + A256(in_b) + A256(out_b) + A256(w_b));
    int rc = mhip_h2d_async(din, input, in_b);
    if (!rc) rc = mhip_h2d_async(dw, weight, w_b);
    if (!rc && bias) rc = mhip_h2d_async(db, bias, b_b);
    mhip_conv_f32_t p;
    memset(&p, 0, sizeof(p));
    p.in = din; p.out = dout; p.w = dw; p.bias = bias ? db : NULL; p.frames = 1;
    p.in_h = in_h; p.in_w = in_w; p.in_c = in_c; p.out_h = out_h; p.out_w = out_w; p.out_c = out_c;
    p.kh = kh; p.kw = kw; p.stride_h = stride_h; p.stride_w = stride_w; p.pad_top = pad_top; p.pad_left = pad_left;
    if (!rc) rc = mhip_conv_f32(&p);
    if (!rc) rc = mhip_d2h_async(output, dout, out_b);
    if (mhip_sync() || rc) fprintf(stderr, "%s: GPU execution failed: %s\n", who, mhip_last_error());
    mhip_free(d);
}
