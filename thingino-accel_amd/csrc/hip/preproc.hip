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

// Tiled form: a workgroup owns a 16 x 16 output tile.  The source region the tile's gather lists touch is staged
// once in LDS (coalesced 4-byte loads), the horizontal pass is evaluated once per (region row, output column,
// channel) into an LDS float buffer, the vertical pass reads that buffer: the same float operations in the same
// order as the direct kernel (and the reference), ~taps-fold fewer of them and no scattered byte loads from HBM.
// Used whenever region + buffer fit the LDS budget the host computed (max_cols, max_rows).
__global__ __launch_bounds__(256) void letterbox_tiled_kernel(const mhip_letterbox_t p, const int rcols_max, const int rrows_max) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    float *dec = (float *)sm;                                   // [256]
    float *hbuf = dec + 256;                                    // [rrows_max][16][3]
    unsigned char *reg = (unsigned char *)(hbuf + rrows_max * 48); // [rrows_max][row_bytes]
    const int row_bytes = (rcols_max * 3 + 3) & ~3;
    const int tid = threadIdx.x;
    dec[tid] = (float)tid / 255.0f;
    const int x = blockIdx.x * 16 + (tid & 15), y = blockIdx.y * 16 + (tid >> 4);
    const int f = blockIdx.z;
    const uint8_t *src = p.rgb + (size_t)f * p.rgb_stride;
    int8_t *dst = p.out + (size_t)f * p.out_stride;
    // the part of this tile that lies inside the resized image
    const int rx0 = max((int)blockIdx.x * 16 - p.px, 0), rx1 = min((int)blockIdx.x * 16 + 16 - p.px, p.nw);
    const int ry0 = max((int)blockIdx.y * 16 - p.py, 0), ry1 = min((int)blockIdx.y * 16 + 16 - p.py, p.nh);
    const bool any = rx1 > rx0 && ry1 > ry0; // uniform
    int xs0 = 0, ys0 = 0;
    if (any) {
        // extent of the sources this tile touches (windows of neighbouring outputs are not strictly ordered once
        // zero weights are dropped, so take min / max over the tile's outputs)
        int xs1 = 0, ys1 = 0;
        xs0 = ys0 = 0x7fffffff;
        for (int o = rx0; o < rx1; o++) {
            xs0 = min(xs0, p.xsrc[p.xstart[o]]);
            xs1 = max(xs1, p.xsrc[p.xstart[o + 1] - 1]);
        }
        for (int o = ry0; o < ry1; o++) {
            ys0 = min(ys0, p.ysrc[p.ystart[o]]);
            ys1 = max(ys1, p.ysrc[p.ystart[o + 1] - 1]);
        }
        const int ncols = xs1 - xs0 + 1, nrows = ys1 - ys0 + 1;
        const int nb = ncols * 3;
        // stage the region: row r = bytes [xs0*3, xs0*3 + nb) of source row ys0 + r
        const int dwords = (nb + 3) >> 2;
        for (int i = tid; i < nrows * dwords; i += 256) {
            const int r = i / dwords, d = i - r * dwords;
            const uint8_t *g = src + ((size_t)(ys0 + r) * p.w + xs0) * 3 + d * 4;
            uint32_t v = 0;
            if (d * 4 + 4 <= nb) __builtin_memcpy(&v, g, 4); // unaligned dword load (any alignment on gfx950)
            else
                for (int b = 0; b < nb - d * 4; b++) v |= (uint32_t)g[b] << (8 * b);
            *(uint32_t *)(reg + r * row_bytes + d * 4) = v;
        }
        __syncthreads();
        // horizontal pass: one (region row, tile column) per work item, all three channels
        const int ncx = rx1 - rx0;
        for (int i = tid; i < nrows * ncx; i += 256) {
            const int r = i / ncx, cx = i - r * ncx, rx = rx0 + cx;
            const unsigned char *row = reg + r * row_bytes;
            float h0 = 0.0f, h1 = 0.0f, h2 = 0.0f;
            for (int k = p.xstart[rx]; k < p.xstart[rx + 1]; k++) {
                const unsigned char *q = row + (p.xsrc[k] - xs0) * 3;
                const float wk = p.xw[k];
                h0 = h0 + dec[q[0]] * wk;
                h1 = h1 + dec[q[1]] * wk;
                h2 = h2 + dec[q[2]] * wk;
            }
            float *hb = hbuf + (r * 16 + cx) * 3;
            hb[0] = h0; hb[1] = h1; hb[2] = h2;
        }
    }
    __syncthreads();
    if (x >= p.tw || y >= p.th) return;
    int v8[3] = {-17, -17, -17};
    const int rx = x - p.px, ry = y - p.py;
    if (rx >= 0 && rx < p.nw && ry >= 0 && ry < p.nh) {
        float acc[3] = {0.0f, 0.0f, 0.0f};
        const int cx = rx - rx0;
        for (int j = p.ystart[ry]; j < p.ystart[ry + 1]; j++) {
            const float *hb = hbuf + ((p.ysrc[j] - ys0) * 16 + cx) * 3;
            const float wj = p.yw[j];
#pragma unroll
            for (int c = 0; c < 3; c++) acc[c] = acc[c] + hb[c] * wj;
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
    if (p->max_cols > 0 && p->max_rows > 0) {
        const size_t lds = 1024 + (size_t)p->max_rows * 48 * 4 + (size_t)p->max_rows * (((size_t)p->max_cols * 3 + 3) & ~(size_t)3);
        if (lds <= 60 * 1024) {
            hipLaunchKernelGGL(letterbox_tiled_kernel, grid, dim3(256), lds, mhip_stream_native(), *p, p->max_cols, p->max_rows);
            return mhip_check(hipGetLastError(), "letterbox (tiled)");
        }
    }
    hipLaunchKernelGGL(letterbox_kernel, grid, dim3(256), 0, mhip_stream_native(), *p);
    return mhip_check(hipGetLastError(), "letterbox");
}
