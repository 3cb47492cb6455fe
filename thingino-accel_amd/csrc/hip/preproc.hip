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

// Strip form (round 6).  The tiled form above gave a workgroup one 16 x 16 output tile: 409 600 workgroups per
// 256-frame batch, every one reading its gather lists from global memory, looking every source byte up once PER TAP, and storing three
// single bytes per thread -- 4.5 ms per batch = 228 GB/s of the 1 GB it moves, as long as the whole 60-layer graph.  Here a workgroup owns
// LB_R output rows over the full width and STREAMS the source rows its vertical lists touch, in increasing order:
//   * the horizontal gather list (start / source / weight) is staged in LDS once per workgroup;
//   * a source row arrives by 16-byte loads (the next row's loads are in flight while this one is used) and is converted to floats
//     ONCE per byte (the /255 table) into an LDS row -- taps then read floats, not bytes through a table;
//   * a thread owns NC output columns (x = tid + 256 i): it evaluates the row's horizontal sums for them (h = h + v * w, increasing
//     source order from 0: the library's order) and adds them into the accumulators of those of its LB_R output rows whose vertical list
//     holds this source row (acc = acc + h * w, increasing source order: each output's list is walked by a pointer that only moves on).
//     Which rows those are is the same for every thread (scalar control flow);
//   * the strip's bytes are put together in LDS and leave as 16-byte stores of contiguous runs (NHWC rows are 3 * tw contiguous bytes).
// The same float operations in the same order as the two kernels above, bit for bit (tests/test_gpu_preproc.py); letterbox bands
// (strips outside the resized image) are filled without touching the source.
#define LB_R 8
#define LB_YT 64 // entries per output row of the staged vertical lists (longer lists: the launcher takes another form)
template <int NC>
__global__ __launch_bounds__(256) void letterbox_strip_kernel(const mhip_letterbox_t p, const int n_xtaps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x;
    const int row_f = ((p.w * 3 + 15) & ~15);                 // floats per staged source row (the last dword's tail lands here too)
    float *dec = (float *)sm;                                 // [256]
    int *xs_l = (int *)(dec + 256);                           // [nw + 1] (+ pad to a multiple of 4)
    int2 *tap_l = (int2 *)(xs_l + ((p.nw + 1 + 3) & ~3));     // [n_xtaps] {source column * 3, weight bits}
    int2 *yl = tap_l + ((n_xtaps + 1) & ~1);                  // [LB_R][LB_YT] {source row, weight bits} of the strip's rows, -1 terminated
    float *rowf = (float *)(yl + LB_R * LB_YT);               // [row_f]: ONE float row (two of them put the workgroup at 84 KB: one workgroup, i.e. one wave per SIMD, per CU)
    int8_t *stage = (int8_t *)rowf;                           // (after the last source row) [LB_R][tw * 3] output bytes
    const int f = blockIdx.y, y0 = blockIdx.x * LB_R;
    const uint8_t *src = p.rgb + (size_t)f * p.rgb_stride;
    int8_t *dst = p.out + (size_t)f * p.out_stride;
    const int rows = min(LB_R, p.th - y0);
    const int row_b = p.tw * 3;
    // the strip's rows inside the resized image: [r_lo, r_hi) of its LB_R
    const int r_lo = max(p.py - y0, 0), r_hi = min(p.py + p.nh - y0, rows);
    if (r_hi <= r_lo) { // a letterbox band: grey, 16 bytes per lane (NHWC and planar alike: every byte of these rows is -17)
        const unsigned g = 0xefefefefu; // (int8) -17
        if (p.nhwc) {
            int8_t *o = dst + (size_t)y0 * row_b;
            const int n = rows * row_b;
            if ((((uintptr_t)o | (unsigned)n) & 15) == 0) {
                for (int i = tid * 16; i < n; i += 256 * 16) *(uint4 *)(o + i) = make_uint4(g, g, g, g);
            } else {
                for (int i = tid; i < n; i += 256) o[i] = (int8_t)-17;
            }
        } else {
            for (int c = 0; c < 3; c++) {
                int8_t *o = dst + ((size_t)c * p.th + y0) * p.tw;
                for (int i = tid; i < rows * p.tw; i += 256) o[i] = (int8_t)-17;
            }
        }
        return;
    }
    dec[tid] = (float)tid / 255.0f;
    for (int i = tid; i <= p.nw; i += 256) xs_l[i] = p.xstart[i];
    for (int i = tid; i < n_xtaps; i += 256) tap_l[i] = make_int2(p.xsrc[i] * 3, __float_as_int(p.xw[i]));
    // vertical lists of the strip's rows: staged in LDS once (walking them in global memory cost a dependent scalar load per output row and
    // source row: ~2400 cycles of latency per source row), each behind a terminator no source row matches; jp[r] = the list's read pointer
    int jp[LB_R];
    int s0 = 0x7fffffff, s1 = -1;
#pragma unroll
    for (int r = 0; r < LB_R; r++) {
        const bool in = r >= r_lo && r < r_hi;
        const int ry = in ? y0 + r - p.py : 0;
        const int j0 = in ? p.ystart[ry] : 0, j1 = in ? p.ystart[ry + 1] : 0;
        jp[r] = r * LB_YT;
        for (int j = j0 + tid; j < j1; j += 256) yl[r * LB_YT + j - j0] = make_int2(p.ysrc[j], __float_as_int(p.yw[j]));
        if (tid == 0) yl[r * LB_YT + j1 - j0] = make_int2(-1, 0);
        if (j1 > j0) {
            s0 = min(s0, p.ysrc[j0]);
            s1 = max(s1, p.ysrc[j1 - 1]);
        }
    }
    float acc[LB_R][NC][3];
#pragma unroll
    for (int r = 0; r < LB_R; r++)
#pragma unroll
        for (int i = 0; i < NC; i++) acc[r][i][0] = acc[r][i][1] = acc[r][i][2] = 0.0f;
    // a source row = w * 3 bytes from a 1-byte-aligned address, fetched as dwords (unaligned ones are served on gfx950): dword d of the row
    // belongs to thread d % 256, so that a wave reads 256 contiguous bytes and -- converted -- writes 64 consecutive float4s (lane-linear:
    // no bank conflicts; 16 bytes per thread put the lanes' float4s 64 bytes apart, a 4-way conflict on every write).  A ragged last dword
    // is put together from single bytes: nothing is read beyond the row
    const int rb = p.w * 3, nld = (rb + 3) >> 2; // dwords per row (at most 8 per thread: the launcher checks)
    unsigned ld[8];
    auto fetch = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = tid + 256 * k;
            if (i < nld) {
                const uint8_t *g = src + (size_t)s * rb + (size_t)i * 4;
                if (i * 4 + 4 <= rb) __builtin_memcpy(&ld[k], g, 4);
                else {
                    unsigned v = 0u;
                    for (int b = 0; b < rb - i * 4; b++) v |= (unsigned)g[b] << (8 * b);
                    ld[k] = v;
                }
            }
        }
    };
    if (s1 >= s0) fetch(s0);
    for (int s = s0; s <= s1; s++) {
        float *rf = rowf;
#pragma unroll
        for (int k = 0; k < 8; k++) { // bytes -> floats, once per byte
            const int i = tid + 256 * k;
            if (i < nld) {
                const unsigned v = ld[k];
                *(float4 *)(rf + i * 4) = make_float4(dec[v & 255u], dec[(v >> 8) & 255u], dec[(v >> 16) & 255u], dec[v >> 24]);
            }
        }
        if (s < s1) fetch(s + 1);
        __syncthreads(); // (also orders the tables / dec staged above before their first use)
        // the heads of the eight vertical lists: one batch of (broadcast) LDS reads; does any of the strip's rows use this source row?
        int2 head[LB_R];
        bool used = false;
#pragma unroll
        for (int r = 0; r < LB_R; r++) {
            head[r] = yl[jp[r]];
            head[r].x = __builtin_amdgcn_readfirstlane(head[r].x);
            head[r].y = __builtin_amdgcn_readfirstlane(head[r].y);
            used |= head[r].x == s;
        }
        if (!used) {
            __syncthreads(); // (pairs with the barrier behind the reads below: the next row's floats overwrite this one's)
            continue;
        }
        // horizontal sums of this thread's NC columns, their lists walked SIDE BY SIDE (3 * NC independent chains and NC tap / pixel reads in
        // flight per round instead of one column after the other), the next round's taps read ahead of this round's additions
        float h[NC][3];
        int kk[NC], kn[NC], nmax = 0;
#pragma unroll
        for (int i = 0; i < NC; i++) {
            const int x = tid + 256 * i;
            h[i][0] = h[i][1] = h[i][2] = 0.0f;
            kk[i] = x < p.nw ? xs_l[x] : 0;
            kn[i] = x < p.nw ? xs_l[x + 1] - kk[i] : 0;
            nmax = max(nmax, kn[i]);
        }
        for (int j = 0; j < nmax; j++) {
            int2 t[NC];
            float v[NC][3];
#pragma unroll
            for (int i = 0; i < NC; i++) {
                t[i] = j < kn[i] ? tap_l[kk[i] + j] : make_int2(0, 0); // (past a column's list: weight +0, pixel 0 -- not added, see below)
                const float *q = rf + t[i].x;
                v[i][0] = q[0]; v[i][1] = q[1]; v[i][2] = q[2];
            }
#pragma unroll
            for (int i = 0; i < NC; i++)
                if (j < kn[i]) { // (h + x * (+0) would turn a -0 sum into +0 and an inf pixel into NaN: entries past the list are not added)
                    const float wk = __int_as_float(t[i].y);
                    h[i][0] = h[i][0] + v[i][0] * wk;
                    h[i][1] = h[i][1] + v[i][1] * wk;
                    h[i][2] = h[i][2] + v[i][2] * wk;
                }
        }
#pragma unroll
        for (int r = 0; r < LB_R; r++) {
            while (head[r].x == s) { // (clamped edges repeat a source row: every entry is its own addition, in list order)
                const float wj = __int_as_float(head[r].y);
#pragma unroll
                for (int i = 0; i < NC; i++) {
                    acc[r][i][0] = acc[r][i][0] + h[i][0] * wj;
                    acc[r][i][1] = acc[r][i][1] + h[i][1] * wj;
                    acc[r][i][2] = acc[r][i][2] + h[i][2] * wj;
                }
                jp[r]++;
                head[r] = yl[jp[r]];
                head[r].x = __builtin_amdgcn_readfirstlane(head[r].x);
                head[r].y = __builtin_amdgcn_readfirstlane(head[r].y);
            }
        }
        __syncthreads(); // every thread has read this row's floats: the next row may take their place
    }
    __syncthreads(); // every thread is done with the float rows: the output bytes take their place
    // grey everywhere first (bands left / right of the image, rows of the strip outside it), then the computed pixels
    const bool planar = !p.nhwc;
    const int nst = rows * row_b;
    for (int i = tid * 4; i < nst; i += 1024) *(unsigned *)(stage + i) = 0xefefefefu; // (nst % 4 == 0: row_b = 3 tw, checked by the launcher)
    __syncthreads();
#pragma unroll
    for (int r = 0; r < LB_R; r++) {
        if (r < r_lo || r >= r_hi) continue;
#pragma unroll
        for (int i = 0; i < NC; i++) {
            const int x = tid + 256 * i;
            if (x >= p.nw) continue;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float v = acc[r][i][c];
                v = v < 0.0f ? 0.0f : v;
                v = v > 1.0f ? 1.0f : v;
                const float t = v * 255.0f;
                const int q = (int)((double)t + 0.5);
                const int8_t b = (int8_t)((unsigned char)q - 128);
                if (planar) stage[(r * 3 + c) * p.tw + p.px + x] = b;
                else stage[r * row_b + (p.px + x) * 3 + c] = b;
            }
        }
    }
    __syncthreads();
    if (!planar) { // rows y0 .. y0 + rows - 1 are one contiguous run of the frame
        int8_t *o = dst + (size_t)y0 * row_b;
        if ((((uintptr_t)o | (unsigned)nst) & 15) == 0) {
            for (int i = tid * 16; i < nst; i += 256 * 16) *(uint4 *)(o + i) = *(const uint4 *)(stage + i);
        } else {
            for (int i = tid; i < nst; i += 256) o[i] = stage[i];
        }
    } else { // [3][th][tw]: per channel, rows y0 .. are contiguous
        for (int c = 0; c < 3; c++) {
            int8_t *o = dst + ((size_t)c * p.th + y0) * p.tw;
            for (int i = tid; i < rows * p.tw; i += 256) o[i] = stage[((i / p.tw) * 3 + c) * p.tw + i % p.tw];
        }
    }
}

static size_t strip_lds(const mhip_letterbox_t *p) {
    const size_t row_f = ((size_t)p->w * 3 + 15) & ~(size_t)15;
    return 1024 + (((size_t)p->nw + 1 + 3) & ~(size_t)3) * 4 + (((size_t)p->n_xtaps + 1) & ~(size_t)1) * 8 + (size_t)LB_R * LB_YT * 8 + row_f * 4;
}

extern "C" int mhip_letterbox(const mhip_letterbox_t *p) {
    if (!p || !p->rgb || !p->out || !p->xstart || !p->xsrc || !p->xw || !p->ystart || !p->ysrc || !p->yw) return -1;
    if (p->frames <= 0 || p->w <= 0 || p->h <= 0 || p->tw <= 0 || p->th <= 0 || p->nw <= 0 || p->nh <= 0 || p->px < 0 ||
        p->py < 0 || p->px + p->nw > p->tw || p->py + p->nh > p->th || p->frames > 65535)
        return -1;
    // strip form: the gather list and two float rows fit LDS, a source row is at most 512 16-byte pieces, at most 4 columns per thread,
    // and the strip's output bytes fit where the float rows were
    if (p->n_xtaps > 0 && p->max_ytaps > 0 && p->max_ytaps < LB_YT && p->form != 1 && p->form != 2 && p->nw <= 1024 && (p->w * 3 + 3) / 4 <= 2048 && (p->tw * 3) % 4 == 0 && strip_lds(p) <= 80 * 1024 &&
        (size_t)LB_R * p->tw * 3 <= ((((size_t)p->w * 3 + 15) & ~(size_t)15) * 4)) {
        const dim3 g((unsigned)((p->th + LB_R - 1) / LB_R), (unsigned)p->frames);
        const size_t lds = strip_lds(p);
        const int nc = (p->nw + 255) / 256;
#define LB_LAUNCH(NC)                                                                                                                     \
    do {                                                                                                                                  \
        static bool attr_##NC = false;                                                                                                    \
        if (!attr_##NC) {                                                                                                                 \
            if (hipFuncSetAttribute((const void *)letterbox_strip_kernel<NC>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) \
                return mhip_check(hipErrorUnknown, "letterbox (strips) attribute");                                                       \
            attr_##NC = true;                                                                                                             \
        }                                                                                                                                 \
        hipLaunchKernelGGL(letterbox_strip_kernel<NC>, g, dim3(256), lds, mhip_stream_native(), *p, p->n_xtaps);                          \
    } while (0)
        if (nc <= 1) LB_LAUNCH(1);
        else if (nc == 2) LB_LAUNCH(2);
        else if (nc == 3) LB_LAUNCH(3);
        else LB_LAUNCH(4);
#undef LB_LAUNCH
        return mhip_check(hipGetLastError(), "letterbox (strips)");
    }
    dim3 grid((unsigned)((p->tw + 15) / 16), (unsigned)((p->th + 15) / 16), (unsigned)p->frames);
    if (p->max_cols > 0 && p->max_rows > 0 && p->form != 2) {
        const size_t lds = 1024 + (size_t)p->max_rows * 48 * 4 + (size_t)p->max_rows * (((size_t)p->max_cols * 3 + 3) & ~(size_t)3);
        if (lds <= 60 * 1024) {
            hipLaunchKernelGGL(letterbox_tiled_kernel, grid, dim3(256), lds, mhip_stream_native(), *p, p->max_cols, p->max_rows);
            return mhip_check(hipGetLastError(), "letterbox (tiled)");
        }
    }
    hipLaunchKernelGGL(letterbox_kernel, grid, dim3(256), 0, mhip_stream_native(), *p);
    return mhip_check(hipGetLastError(), "letterbox");
}
