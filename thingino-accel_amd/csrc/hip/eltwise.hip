// eltwise.hip -- element-wise layers of the .mars executor on gfx950.
// HBM-bound byte/float streaming: 16 bytes per lane, batch on grid.y.
//
// Replaces the scalar loops of reference src/mars/mars_runtime.c:
//   :700-707 fused-ReLU byte pass, :752-768 int8 sigmoid (here: 256-entry LUT
//   built on the host with the host's libm, so bit-exact by construction; the f32
//   sigmoid uses expf_exact.h, a restatement of glibc's expf that matches it bit for bit),
//   :818-835 / :885-902 int8 mul/add, :742-749 / :807-816 / :874-883 f32 forms,
//   :1066-1086 relu / leaky relu, :1115-1154 batchnorm.
// Float arithmetic is written so that every operation rounds once, in the
// order of the reference (compile with -ffp-contract=off).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../expf_exact.h"
#include "../mhip.h"

extern "C" hipStream_t mhip_stream_native(void);
extern "C" int mhip_check(hipError_t e, const char *what);

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define EW_THREADS 256

static inline dim3 ew_grid(size_t chunks, int frames) {
    size_t bx = (chunks + EW_THREADS - 1) / EW_THREADS;
    if (bx == 0) bx = 1;
    return dim3((unsigned)bx, (unsigned)frames);
}

__device__ __forceinline__ int trunc_x86(float v) {
    int r = (int)v;
    if (!(v < 2147483648.0f)) r = INT_MIN;
    return r;
}
__device__ __forceinline__ int sat8(int v) { return v > 127 ? 127 : (v < -128 ? -128 : v); }

// ------------------------------------------------------------------ LUT map
__global__ __launch_bounds__(EW_THREADS) void lut_i8_kernel(const int8_t *in, size_t is, int8_t *out, size_t os,
                                                            size_t n, const uint8_t *lut, int vec) {
    __shared__ uint8_t sl[256];
    sl[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const int8_t *src = in + (size_t)blockIdx.y * is;
    int8_t *dst = out + (size_t)blockIdx.y * os;
    size_t c = (size_t)blockIdx.x * EW_THREADS + threadIdx.x;
    size_t i0 = c * 16;
    if (i0 >= n) return;
    if (vec && i0 + 16 <= n) {
        v4i v = *(const v4i *)(src + i0);
        v4i o;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            uint32_t w = (uint32_t)v[d], r = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                uint32_t q = (w >> (8 * b)) & 255u;
                r |= (uint32_t)sl[(q + 128u) & 255u] << (8 * b); // index = int8 value + 128
            }
            o[d] = (int)r;
        }
        *(v4i *)(dst + i0) = o;
    } else {
        size_t e = i0 + 16 < n ? i0 + 16 : n;
        for (size_t i = i0; i < e; i++) dst[i] = (int8_t)sl[(uint8_t)(src[i] + 128)];
    }
}

extern "C" int mhip_lut_i8(const int8_t *in, size_t in_stride, int8_t *out, size_t out_stride, int frames,
                           size_t n, const uint8_t *lut_dev) {
    if (!in || !out || !lut_dev || frames <= 0) return -1;
    if (n == 0) return 0;
    int vec = (((uintptr_t)in | (uintptr_t)out | in_stride | out_stride) & 15) == 0;
    hipLaunchKernelGGL(lut_i8_kernel, ew_grid((n + 15) / 16, frames), dim3(EW_THREADS), 0, mhip_stream_native(), in,
                       in_stride, out, out_stride, n, lut_dev, vec);
    return mhip_check(hipGetLastError(), "lut_i8");
}

// ------------------------------------------------------- byte ReLU in place
__global__ __launch_bounds__(EW_THREADS) void relu_bytes_kernel(int8_t *buf, size_t stride, size_t n, int vec) {
    int8_t *p = buf + (size_t)blockIdx.y * stride;
    size_t i0 = ((size_t)blockIdx.x * EW_THREADS + threadIdx.x) * 16;
    if (i0 >= n) return;
    if (vec && i0 + 16 <= n) {
        v4i v = *(v4i *)(p + i0);
#pragma unroll
        for (int d = 0; d < 4; d++) {
            uint32_t w = (uint32_t)v[d];
            uint32_t neg = (w & 0x80808080u) >> 7;       // 1 per negative byte
            uint32_t mask = (neg * 255u);                // 0xFF per negative byte
            v[d] = (int)(w & ~mask);
        }
        *(v4i *)(p + i0) = v;
    } else {
        size_t e = i0 + 16 < n ? i0 + 16 : n;
        for (size_t i = i0; i < e; i++)
            if (p[i] < 0) p[i] = 0;
    }
}

extern "C" int mhip_relu_bytes(int8_t *buf, size_t stride, int frames, size_t n) {
    if (!buf || frames <= 0) return -1;
    if (n == 0) return 0;
    int vec = (((uintptr_t)buf | stride) & 15) == 0;
    hipLaunchKernelGGL(relu_bytes_kernel, ew_grid((n + 15) / 16, frames), dim3(EW_THREADS), 0, mhip_stream_native(),
                       buf, stride, n, vec);
    return mhip_check(hipGetLastError(), "relu_bytes");
}

// -------------------------------------------------------------- int8 binary
__device__ __forceinline__ int8_t binary_one(int is_mul, int8_t a, int8_t b, float sa, float sb, float inv) {
    float va = (float)a * sa;
    float vb = (float)b * sb;
    float y = is_mul ? va * vb : va + vb;
    float t = y * inv;
    return (int8_t)sat8(trunc_x86(t + 0.5f));
}

__global__ __launch_bounds__(EW_THREADS) void binary_i8_kernel(int is_mul, const int8_t *a, size_t as, const int8_t *b,
                                                               size_t bs, int8_t *out, size_t os, size_t n, float sa,
                                                               float sb, float inv, int vec, int run, int pstride,
                                                               int choff) {
    const int8_t *pa = a + (size_t)blockIdx.y * as;
    const int8_t *pb = b + (size_t)blockIdx.y * bs;
    int8_t *po = out + (size_t)blockIdx.y * os;
    size_t i0 = ((size_t)blockIdx.x * EW_THREADS + threadIdx.x) * 16;
    if (i0 >= n) return;
    if (run > 0) { // channel-slice output: element i -> pixel i/run, channel i%run of a wider tensor
        const size_t pix = i0 / (size_t)run;
        po += pix * (size_t)pstride + choff - pix * (size_t)run; // vec path: run % 16 == 0, a chunk never crosses a pixel
        if (!vec) {
            size_t e = i0 + 16 < n ? i0 + 16 : n;
            for (size_t i = i0; i < e; i++) {
                const size_t px = i / (size_t)run;
                out[(size_t)blockIdx.y * os + px * (size_t)pstride + choff + (i - px * (size_t)run)] =
                    binary_one(is_mul, pa[i], pb[i], sa, sb, inv);
            }
            return;
        }
    }
    if (vec && i0 + 16 <= n) {
        v4i x = *(const v4i *)(pa + i0), y = *(const v4i *)(pb + i0), o;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            uint32_t r = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int8_t ea = (int8_t)((uint32_t)x[d] >> (8 * k));
                int8_t eb = (int8_t)((uint32_t)y[d] >> (8 * k));
                r |= (uint32_t)(uint8_t)binary_one(is_mul, ea, eb, sa, sb, inv) << (8 * k);
            }
            o[d] = (int)r;
        }
        *(v4i *)(po + i0) = o;
    } else {
        size_t e = i0 + 16 < n ? i0 + 16 : n;
        for (size_t i = i0; i < e; i++) po[i] = binary_one(is_mul, pa[i], pb[i], sa, sb, inv);
    }
}

extern "C" int mhip_binary_i8(int is_mul, const int8_t *a, size_t a_stride, const int8_t *b, size_t b_stride,
                              int8_t *out, size_t out_stride, int frames, size_t n, float sa, float sb,
                              float inv_so, int out_run, int out_pix_stride, int out_ch_off) {
    if (!a || !b || !out || frames <= 0 || out_run < 0 || out_pix_stride < 0 || out_ch_off < 0) return -1;
    if (out_run > 0 && (out_pix_stride < out_ch_off + out_run || n % (size_t)out_run != 0)) return -1;
    if (n == 0) return 0;
    int vec = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out | a_stride | b_stride | out_stride) & 15) == 0;
    if (out_run > 0 && ((out_run | out_pix_stride | out_ch_off) & 15)) vec = 0;
    hipLaunchKernelGGL(binary_i8_kernel, ew_grid((n + 15) / 16, frames), dim3(EW_THREADS), 0, mhip_stream_native(),
                       is_mul, a, a_stride, b, b_stride, out, out_stride, n, sa, sb, inv_so, vec, out_run,
                       out_pix_stride, out_ch_off);
    return mhip_check(hipGetLastError(), "binary_i8");
}

// ---------------------------------------------------------------- f32 forms
// op: 0 add, 1 mul, 2 sub, 3 sigmoid(a), 4 relu/leaky(a, alpha)
__global__ __launch_bounds__(EW_THREADS) void f32_kernel(int op, const float *a, size_t as, const float *b, size_t bs,
                                                         float *out, size_t os, size_t n, float alpha) {
    const float *pa = (const float *)((const char *)a + (size_t)blockIdx.y * as);
    const float *pb = b ? (const float *)((const char *)b + (size_t)blockIdx.y * bs) : nullptr;
    float *po = (float *)((char *)out + (size_t)blockIdx.y * os);
    size_t i0 = ((size_t)blockIdx.x * EW_THREADS + threadIdx.x) * 4;
    for (size_t i = i0; i < i0 + 4 && i < n; i++) {
        float x = pa[i], r;
        switch (op) {
            case 0: r = x + pb[i]; break;
            case 1: r = x * pb[i]; break;
            case 2: r = x - pb[i]; break;
            case 3: r = 1.0f / (1.0f + expf_exact(-x, expf_exact_tab)); break; // libm-exact expf: see expf_exact.h
            default: r = x > 0.0f ? x : x * alpha; break;
        }
        po[i] = r;
    }
}

static int launch_f32(int op, const float *a, size_t as, const float *b, size_t bs, float *out, size_t os,
                      int frames, size_t n, float alpha) {
    if (!a || !out || frames <= 0 || (op <= 2 && !b)) return -1;
    if (n == 0) return 0;
    hipLaunchKernelGGL(f32_kernel, ew_grid((n + 3) / 4, frames), dim3(EW_THREADS), 0, mhip_stream_native(), op, a, as,
                       b, bs, out, os, n, alpha);
    return mhip_check(hipGetLastError(), "f32 eltwise");
}

extern "C" int mhip_sigmoid_f32(const float *in, size_t in_stride, float *out, size_t out_stride, int frames,
                                size_t n) {
    return launch_f32(3, in, in_stride, nullptr, 0, out, out_stride, frames, n, 0.f);
}
extern "C" int mhip_binary_f32(int op, const float *a, size_t a_stride, const float *b, size_t b_stride, float *out,
                               size_t out_stride, int frames, size_t n) {
    if (op < 0 || op > 2) return -1;
    return launch_f32(op, a, a_stride, b, b_stride, out, out_stride, frames, n, 0.f);
}
extern "C" int mhip_relu_f32(const float *in, size_t in_stride, float *out, size_t out_stride, int frames, size_t n,
                             float alpha) {
    return launch_f32(4, in, in_stride, nullptr, 0, out, out_stride, frames, n, alpha);
}

// ---------------------------------------------------------------- batchnorm
// layout per frame: [n][c][hw] (the reference assumes NCHW here, :1104-1108)
__global__ __launch_bounds__(EW_THREADS) void bn_kernel(int is_f32, const void *in, size_t is, void *out, size_t os,
                                                        size_t total, int c, int hw, const float *s, const float *b,
                                                        float in_scale, float out_scale) {
    size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x;
    if (i >= total) return;
    int ci = (int)((i / (size_t)hw) % (size_t)c);
    float sc = s ? s[ci] : 1.0f, bi = b ? b[ci] : 0.0f;
    if (is_f32) {
        const float *pi = (const float *)((const char *)in + (size_t)blockIdx.y * is);
        float *po = (float *)((char *)out + (size_t)blockIdx.y * os);
        float m = pi[i] * sc;
        po[i] = m + bi;
    } else {
        const int8_t *pi = (const int8_t *)in + (size_t)blockIdx.y * is;
        int8_t *po = (int8_t *)out + (size_t)blockIdx.y * os;
        float x = (float)pi[i] * in_scale;
        float m = x * sc;
        float y = m + bi;
        float t = y / out_scale;
        po[i] = (int8_t)sat8(trunc_x86(t + 0.5f));
    }
}

extern "C" int mhip_batchnorm_i8(const int8_t *in, size_t in_stride, int8_t *out, size_t out_stride, int frames, int n,
                                 int c, int hw, const float *s, const float *b, float in_scale, float out_scale) {
    if (!in || !out || frames <= 0 || n <= 0 || c <= 0 || hw <= 0) return -1;
    size_t total = (size_t)n * c * hw;
    hipLaunchKernelGGL(bn_kernel, ew_grid(total, frames), dim3(EW_THREADS), 0, mhip_stream_native(), 0, in, in_stride,
                       out, out_stride, total, c, hw, s, b, in_scale, out_scale);
    return mhip_check(hipGetLastError(), "bn_i8");
}
extern "C" int mhip_batchnorm_f32(const float *in, size_t in_stride, float *out, size_t out_stride, int frames, int n,
                                  int c, int hw, const float *s, const float *b) {
    if (!in || !out || frames <= 0 || n <= 0 || c <= 0 || hw <= 0) return -1;
    size_t total = (size_t)n * c * hw;
    hipLaunchKernelGGL(bn_kernel, ew_grid(total, frames), dim3(EW_THREADS), 0, mhip_stream_native(), 1, in, in_stride,
                       out, out_stride, total, c, hw, s, b, 1.f, 1.f);
    return mhip_check(hipGetLastError(), "bn_f32");
}
