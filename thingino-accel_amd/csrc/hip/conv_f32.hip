// conv_f32.hip -- float32 convolution, NCHW / OIHW, for gfx950.
//
// Replaces reference src/mars/mxu_conv.c:673-710 (conv2d_float32_mxu; scalar
// even on the camera).  The reference accumulates sequentially
//     sum = bias; for ic, kh, kw (in-image taps only): sum += in * w
// with one rounded multiply and one rounded add per tap.  This kernel keeps
// that exact order and those exact roundings per output element (no FMA
// contraction: built with -ffp-contract=off), so it is bit-identical to the
// reference rather than merely within the 1e-4 the task allows.  One lane per
// output element; a wave covers 64 consecutive pixels of one output channel, so the
// weight stream is wave-uniform (scalar loads) and input reads coalesce.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../mhip.h"

extern "C" hipStream_t mhip_stream_native(void);
extern "C" int mhip_check(hipError_t e, const char *what);

__global__ __launch_bounds__(256) void conv_f32_kernel(const mhip_conv_f32_t p) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= p.out_h * p.out_w) return;
    const int oy = pix / p.out_w, ox = pix - oy * p.out_w;
    const int oc = blockIdx.y;
    const int f = blockIdx.z;
    const float *in = (const float *)((const char *)p.in + (size_t)f * p.in_stride);
    float *out = (float *)((char *)p.out + (size_t)f * p.out_stride);
    const float *wk = p.w + (size_t)oc * p.in_c * p.kh * p.kw;
    const size_t plane = (size_t)p.in_h * p.in_w;
    float acc = p.bias ? p.bias[oc] : 0.0f;
    const int y0 = oy * p.stride_h - p.pad_top, x0 = ox * p.stride_w - p.pad_left;
    for (int ic = 0; ic < p.in_c; ic++) {
        const float *pl = in + ic * plane;
        for (int ky = 0; ky < p.kh; ky++) {
            const int iy = y0 + ky;
            for (int kx = 0; kx < p.kw; kx++) {
                const int ix = x0 + kx;
                if (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) {
                    float prod = pl[(size_t)iy * p.in_w + ix] * wk[(ic * p.kh + ky) * p.kw + kx];
                    acc = acc + prod;
                }
            }
        }
    }
    out[((size_t)oc * p.out_h + oy) * p.out_w + ox] = acc;
}

extern "C" int mhip_conv_f32(const mhip_conv_f32_t *p) {
    if (!p || !p->in || !p->out || !p->w) return -1;
    if (p->frames <= 0 || p->in_h <= 0 || p->in_w <= 0 || p->in_c <= 0 || p->out_h <= 0 || p->out_w <= 0 ||
        p->out_c <= 0 || p->kh <= 0 || p->kw <= 0 || p->stride_h <= 0 || p->stride_w <= 0)
        return -1;
    if (p->out_c > 65535 || p->frames > 65535) return -1;
    dim3 grid((unsigned)(((long)p->out_h * p->out_w + 255) / 256), (unsigned)p->out_c, (unsigned)p->frames);
    hipLaunchKernelGGL(conv_f32_kernel, grid, dim3(256), 0, mhip_stream_native(), *p);
    return mhip_check(hipGetLastError(), "conv_f32");
}
