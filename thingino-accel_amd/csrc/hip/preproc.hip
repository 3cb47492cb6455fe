// preproc.hip -- image front-end on gfx950: letterbox resize of uint8 RGB frames + (px - 128) int8 conversion.
//
// Replaces reference src/mars/mars_yolo_test.c:40-77 (load_image, after the file decode).  The reference resizes
// with stbir_resize_uint8() of the stb_image_resize header it vendors; the per-axis coefficient tables of that
// library are built on the host (csrc/host/mars_preproc.c) as gather lists -- for every output column / row the
// source indices (clamped, increasing) and float weights -- and this kernel only runs the two accumulation
// passes with the library's exact float steps (no FMA contraction: built with -ffp-contract=off):
//   h(row, x, c) = sum_k (in[row][srcx_k][c] / 255) * wx_k        (float, in increasing source order, from 0)
//   v(x, y, c)   = sum_j h(srcy_j, x, c) * wy_j
//   out          = (uint8)(int)((double)(clamp(v, 0, 1) * 255.0f) + 0.5)  - 128, pad = -17
// One thread per output pixel and frame; the horizontal sums are recomputed for every output row that uses them
// (at most 4/scale rows): bit-identical by construction and far below the cost of reading the frame.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../mhip.h"

extern "C" hipStream_t mhip_stream_native(void);
extern "C" int mhip_check(hipError_t e, const char *what);

__global__ __launch_bounds__(256) void letterbox_kernel(const mhip_letterbox_t p) {
    __shared__ float dec[256]; // value / 255 (IEEE division, once per workgroup instead of once per tap)
    dec[threadIdx.x] = (float)threadIdx.x / 255.0f;
    __syncthreads();
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= p.tw || y >= p.th) return;
    const int f = blockIdx.z;
    const uint8_t *src = p.rgb + (size_t)f * p.rgb_stride;
    int8_t *dst = p.out + (size_t)f * p.out_stride;
    int v8[3] = {-17, -17, -17}; // the reference's grey letterbox (:57)
    const int rx = x - p.px, ry = y - p.py;
    if (rx >= 0 && rx < p.nw && ry >= 0 && ry < p.nh) {
        float acc[3] = {0.0f, 0.0f, 0.0f};
        const int x0 = p.xstart[rx], x1 = p.xstart[rx + 1];
        for (int j = p.ystart[ry]; j < p.ystart[ry + 1]; j++) {
            const uint8_t *row = src + (size_t)p.ysrc[j] * p.w * 3;
            float h[3] = {0.0f, 0.0f, 0.0f};
            for (int k = x0; k < x1; k++) {
                const uint8_t *q = row + p.xsrc[k] * 3;
                const float wk = p.xw[k];
#pragma unroll
                for (int c = 0; c < 3; c++) h[c] = h[c] + dec[q[c]] * wk;
            }
            const float wj = p.yw[j];
#pragma unroll
            for (int c = 0; c < 3; c++) acc[c] = acc[c] + h[c] * wj;
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float v = acc[c];
            v = v < 0.0f ? 0.0f : v;
            v = v > 1.0f ? 1.0f : v;
            const float t = v * 255.0f;
            const int r = (int)((double)t + 0.5);
            v8[c] = (int)(int8_t)((unsigned char)r - 128);
        }
    }
    if (p.nhwc) {
        int8_t *o = dst + ((size_t)y * p.tw + x) * 3;
        o[0] = (int8_t)v8[0]; o[1] = (int8_t)v8[1]; o[2] = (int8_t)v8[2];
    } else {
        const size_t ps = (size_t)p.tw * p.th, o = (size_t)y * p.tw + x;
        dst[o] = (int8_t)v8[0]; dst[ps + o] = (int8_t)v8[1]; dst[2 * ps + o] = (int8_t)v8[2];
    }
}

extern "C" int mhip_letterbox(const mhip_letterbox_t *p) {
    if (!p || !p->rgb || !p->out || !p->xstart || !p->xsrc || !p->xw || !p->ystart || !p->ysrc || !p->yw) return -1;
    if (p->frames <= 0 || p->w <= 0 || p->h <= 0 || p->tw <= 0 || p->th <= 0 || p->nw <= 0 || p->nh <= 0 || p->px < 0 ||
        p->py < 0 || p->px + p->nw > p->tw || p->py + p->nh > p->th || p->frames > 65535)
        return -1;
    dim3 grid((unsigned)((p->tw + 15) / 16), (unsigned)((p->th + 15) / 16), (unsigned)p->frames);
    hipLaunchKernelGGL(letterbox_kernel, grid, dim3(256), 0, mhip_stream_native(), *p);
    return mhip_check(hipGetLastError(), "letterbox");
}
