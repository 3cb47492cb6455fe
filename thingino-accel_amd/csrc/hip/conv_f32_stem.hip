// conv_f32_stem.hip -- the float32 twins' first layer (NCHW / OIHW, 3 input channels, 6 x 6, stride 2; reference
// src/mars/mxu_conv.c:673-710) on the bf16 matrix cores with split operands (use_mfma == 3, three piece products: conv_f32_split.hip
// explains the arithmetic), the input patch of a 16 x 32 output tile staged and split ONCE.  Round 5.
//
// Why its own kernel.  conv_f32_split gathers every input element 18 times through L1 for this layer (2.9 ms at batch 256, its
// stamps: 4600 of 8100 cycles per K step in the fetch), and conv_f32_patch's chunks are 8 channels deep.  Here
//   * K order = (kernel row ky, column pair j, column parity, channel slot): a UNIT (8 K-elements, 16 bytes) is the 4 channel slots
//     (3 real + a zero) of two adjacent input columns.  With stride 2 the pair an output column needs for taps (2j, 2j + 1)
//     starts at input column 2 ox + 2 j - pad: pair-aligned for an even pad, so the patch lives in LDS as 32-byte PAIR RECORDS
//     [4 + 4 x hi | 4 + 4 x mid] and the unit of output pixel (oy, ox) is record (2 oy + ky) * PPW + ox + j -- one 16-byte
//     read, aligned, no im2col, no gather.  6 x 3 = 18 units = 5 MFMA K steps (the last two units are dummies and read a zero record);
//   * the weights (32 channels x 160 K-elements x hi / mid) are loaded into REGISTERS once per workgroup: no weight traffic, no
//     K-step barrier at all;
//   * a persistent workgroup walks tiles; per tile ONE barrier: [matrix work on this tile's patch] and [commit of the next tile's
//     patch into the other slot, loads for the tile after it, the epilogue] -- and the two waves of a SIMD take these two halves in
//     opposite order (waves 4-7 finish the previous tile's epilogue and stage first, then multiply), so matrix and vector work of a
//     SIMD's two waves overlap.
// Bound: HBM -- 4.9 MB in + 13.1 MB out per 640 x 640 frame.  Takes in_c <= 4, out_c <= 32, even kernel width <= 8 and kh * kw / 2 <= 20,
// stride 2, even equal pads, output maps that are multiples of 16 x 32.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../mhip.h"

extern "C" hipStream_t mhip_stream_native(void);
extern "C" int mhip_check(hipError_t e, const char *what);

typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define ST_TH 16
#define ST_TW 32
#define ST_NT 512
#define ST_NS 5  // K steps (4 units each)

struct stem_geom_t {
    int in_c, kh, kw, pad;
    int H_in, W_in, H_out, W_out;
    int PR, NG, PPW, dxp;     // patch rows, 4-column groups per row, pair records per row, pair index of tap column 0 of tile column 0
    int nunits;               // kh * kw / 2 (<= 20)
    int nitems;               // PR * NG (<= 1024: two per thread)
    int slot_bytes, lds_bytes;
    int tiles_x, tiles_y;
    unsigned ntiles, in_bytes, per;
};

__device__ __forceinline__ float stem_silu(float v) {
    const float e = __builtin_amdgcn_exp2f(v * -1.44269504088896341f);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}
// 8 floats -> 4 dwords hi, 4 dwords mid (as conv_f32_patch.hip: round to nearest, exact residual, no residual of a non-finite hi)
__device__ __forceinline__ void stem_split8(const float (&x)[8], v4i &hi, v4i &mid) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const f32x2 v = {x[2 * i], x[2 * i + 1]};
        const int h = __builtin_bit_cast(int, __builtin_convertvector(v, bf16x2));
        const float h0 = __int_as_float(h << 16), h1 = __int_as_float(h & (int)0xffff0000);
        f32x2 r;
        asm("v_sub_f32 %0, %1, %2" : "=v"(r[0]) : "v"(v[0]), "v"(h0));
        asm("v_sub_f32 %0, %1, %2" : "=v"(r[1]) : "v"(v[1]), "v"(h1));
        r[0] = __builtin_isfinite(h0) ? r[0] : 0.0f;
        r[1] = __builtin_isfinite(h1) ? r[1] : 0.0f;
        hi[i] = h;
        mid[i] = __builtin_bit_cast(int, __builtin_convertvector(r, bf16x2));
    }
}

// RECOUT (mhip_conv_f32_t.out_rec): the results leave in record format for the one k x k convolution that reads them (layer 3 of the
// yolov5 twins; conv_f32_split.hip says what a record is): the MFMA operands change places, a lane ends with 4 consecutive channels of
// one pixel, v_permlane16_swap pairs them into 16-byte halves of a record.
template <bool RECOUT>
__global__ __launch_bounds__(ST_NT, 2) void conv_f32_stem(const mhip_conv_f32_t p, const stem_geom_t g, const int8_t *__restrict__ wpl) {
    constexpr int MI = 2, NI = 4; // 32 channels x 64 pixels per wave
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    int8_t *zrec = lds;          // 32 zero bytes: what the dummy units read
    int8_t *slots = lds + 256;   // two patch slots

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fc = lane >> 4;
    const bool late = wv >= 4;
    const unsigned hw = (unsigned)(g.H_out * g.W_out);
    const unsigned plane_bytes = (unsigned)(g.H_in * g.W_in) * 4u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, (int)g.in_bytes, 0x00020000);
    if (tid < 8) ((int *)zrec)[tid] = 0;

    // ---- weights -> registers: B operand of step ks, channel tile a: row a * 16 + fr, unit 4 ks + fc
    bf16x8 wh[ST_NS][MI], wm[ST_NS][MI];
    {
        const size_t rowb = (size_t)ST_NS * 4 * 16, planeb = 32 * rowb;
#pragma unroll
        for (int ks = 0; ks < ST_NS; ks++)
#pragma unroll
            for (int a = 0; a < MI; a++) {
                const int8_t *src = wpl + (size_t)(a * 16 + fr) * rowb + (size_t)(4 * ks + fc) * 16;
                wh[ks][a] = __builtin_bit_cast(bf16x8, *(const v4i *)src);
                wm[ks][a] = __builtin_bit_cast(bf16x8, *(const v4i *)(src + planeb));
            }
    }
    // ---- this lane's unit of every step: byte offset of its pair record relative to the pixel's tap-(0, 0) record; -1 = dummy
    int toffl[ST_NS];
#pragma unroll
    for (int ks = 0; ks < ST_NS; ks++) {
        const int u = 4 * ks + fc, ky = u / (g.kw / 2), j = u - ky * (g.kw / 2);
        toffl[ks] = u < g.nunits ? (ky * g.PPW + j) * 32 : -1;
    }
    // ---- A rows: pixel fr of MFMA tile n of this wave = tile row 2 wv + n / 2, column (n & 1) * 16 + fr
    int pbase[NI];
#pragma unroll
    for (int n = 0; n < NI; n++) pbase[n] = ((2 * (2 * wv + (n >> 1))) * g.PPW + g.dxp + (n & 1) * 16 + fr) * 32;

    // ---- fetch items: (patch row, 4-column group), two per thread at most
    int ir[2], ig[2];
    bool has[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int it = tid + q * ST_NT;
        has[q] = it < g.nitems;
        ir[q] = it / g.NG;
        ig[q] = it - ir[q] * g.NG;
    }
    v4i breg[2][4]; // [item][channel]: 4 columns
    auto tile_origin = [&](unsigned t, unsigned &f, int &oy0, int &ox0) __attribute__((always_inline)) {
        const unsigned tpf = (unsigned)(g.tiles_x * g.tiles_y);
        f = t / tpf;
        const unsigned rem = t - f * tpf, ty = rem / (unsigned)g.tiles_x;
        oy0 = (int)ty * ST_TH;
        ox0 = (int)(rem - ty * (unsigned)g.tiles_x) * ST_TW;
    };
    auto fetch = [&](unsigned t) __attribute__((always_inline)) {
        unsigned f;
        int oy0, ox0;
        tile_origin(t, f, oy0, ox0);
        const int iy0 = 2 * oy0 - g.pad, x_al = 2 * ox0 - g.pad - 2 * g.dxp;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int iy = iy0 + ir[q], x = x_al + 4 * ig[q];
            const bool ok = has[q] && t < g.ntiles && iy >= 0 && iy < g.H_in && x >= 0 && x < g.W_in;
            const unsigned vo = ok ? f * (unsigned)p.in_stride + (unsigned)(iy * g.W_in + x) * 4u : 0xffffffffu;
            unsigned so = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                breg[q][c] = c < g.in_c ? __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, so, 0)) : (v4i){0, 0, 0, 0};
                so += plane_bytes;
            }
        }
    };
    auto commit = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if (has[q]) {
                int8_t *dst = slots + slot * g.slot_bytes + (ir[q] * g.PPW + 2 * ig[q]) * 32;
#pragma unroll
                for (int pr = 0; pr < 2; pr++) { // the column pair (2 pr, 2 pr + 1) of the group
                    const float x[8] = {__int_as_float(breg[q][0][2 * pr]),     __int_as_float(breg[q][1][2 * pr]),     __int_as_float(breg[q][2][2 * pr]),     __int_as_float(breg[q][3][2 * pr]),
                                        __int_as_float(breg[q][0][2 * pr + 1]), __int_as_float(breg[q][1][2 * pr + 1]), __int_as_float(breg[q][2][2 * pr + 1]), __int_as_float(breg[q][3][2 * pr + 1])};
                    v4i hi, mid;
                    stem_split8(x, hi, mid);
                    *(v4i *)(dst + pr * 32) = hi;
                    *(v4i *)(dst + pr * 32 + 16) = mid;
                }
            }
        }
    };

    v4f acc[MI][NI], bias4[MI];
#pragma unroll
    for (int a = 0; a < MI; a++) {
        if (RECOUT) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int oc = a * 16 + fc * 4 + j;
                bias4[a][j] = p.bias && oc < p.out_c ? p.bias[oc] : 0.f;
            }
        } else {
            const int oc = a * 16 + fr;
            const float b = p.bias && oc < p.out_c ? p.bias[oc] : 0.f;
            bias4[a] = (v4f){b, b, b, b};
        }
#pragma unroll
        for (int n = 0; n < NI; n++) acc[a][n] = bias4[a];
    }
    auto compute = [&](int slot) __attribute__((always_inline)) {
        const int8_t *sb = slots + slot * g.slot_bytes;
#pragma unroll
        for (int ks = 0; ks < ST_NS; ks++) {
            bf16x8 xh[NI], xm[NI];
#pragma unroll
            for (int n = 0; n < NI; n++) {
                const int8_t *a = toffl[ks] < 0 ? zrec : sb + pbase[n] + toffl[ks];
                xh[n] = __builtin_bit_cast(bf16x8, *(const v4i *)a);
                xm[n] = __builtin_bit_cast(bf16x8, *(const v4i *)(a + 16));
            }
#pragma unroll
            for (int a = 0; a < MI; a++)
#pragma unroll
                for (int n = 0; n < NI; n++) {
                    if (RECOUT) { // weights are the A operand: rows = channels
                        acc[a][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks][a], xm[n], acc[a][n], 0, 0, 0);
                        acc[a][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[ks][a], xh[n], acc[a][n], 0, 0, 0);
                        acc[a][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks][a], xh[n], acc[a][n], 0, 0, 0);
                        continue;
                    }
                    acc[a][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm[n], wh[ks][a], acc[a][n], 0, 0, 0);
                    acc[a][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[n], wm[ks][a], acc[a][n], 0, 0, 0);
                    acc[a][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[n], wh[ks][a], acc[a][n], 0, 0, 0);
                }
        }
    };
    auto epilogue = [&](unsigned t) __attribute__((always_inline)) { // tile t's results: 4 consecutive pixels of a channel row per store
        unsigned f;
        int oy0, ox0;
        tile_origin(t, f, oy0, ox0);
        if (RECOUT) { // lane = pixel fr of tile n, channels 4 fc .. + 3 of channel tile a; lane rows fc, fc ^ 1 trade pieces (every lane: no branch)
#pragma unroll
            for (int n = 0; n < NI; n++) {
                const unsigned pos = (unsigned)((oy0 + 2 * wv + (n >> 1)) * g.W_out + ox0 + (n & 1) * 16 + fr);
                char *ob = (char *)p.out + (size_t)f * p.out_stride + (size_t)pos * 32u + ((fc & 1) ? 0u : 16u);
#pragma unroll
                for (int a = 0; a < MI; a++) {
                    float x[8];
#pragma unroll
                    for (int j = 0; j < 4; j++) x[j] = p.silu ? stem_silu(acc[a][n][j]) : acc[a][n][j];
#pragma unroll
                    for (int j = 4; j < 8; j++) x[j] = 0.f;
                    v4i hi, mid;
                    stem_split8(x, hi, mid);
                    const auto s0 = __builtin_amdgcn_permlane16_swap((unsigned)mid[0], (unsigned)hi[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap((unsigned)mid[1], (unsigned)hi[1], false, false);
                    const int chunk = 2 * a + (fc >> 1);
                    if (chunk * 8 < p.out_c) *(v4i *)(ob + (size_t)chunk * hw * 32u) = (v4i){(int)s0[0], (int)s1[0], (int)s0[1], (int)s1[1]};
                    acc[a][n] = bias4[a];
                }
            }
            return;
        }
#pragma unroll
        for (int n = 0; n < NI; n++) {
            const unsigned pos = (unsigned)((oy0 + 2 * wv + (n >> 1)) * g.W_out + ox0 + (n & 1) * 16 + fc * 4);
#pragma unroll
            for (int a = 0; a < MI; a++) {
                const int oc = a * 16 + fr;
                if (oc < p.out_c) {
                    v4f r = acc[a][n];
                    if (p.silu) {
#pragma unroll
                        for (int j = 0; j < 4; j++) r[j] = stem_silu(r[j]);
                    }
                    const size_t o = (size_t)f * p.out_stride + ((size_t)oc * hw + pos) * 4u;
                    if (p.add) r += *(const v4f *)((const char *)p.add + o);
                    *(v4f *)((char *)p.out + o) = r;
                }
                acc[a][n] = bias4[a];
            }
        }
    };
    auto barrier_lds = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    const unsigned t0 = blockIdx.x * g.per;
    const unsigned t1 = (blockIdx.x + 1) * g.per < g.ntiles ? (blockIdx.x + 1) * g.per : g.ntiles;
    if (t0 >= t1) return;
    __syncthreads(); // the zero record
    fetch(t0);
    commit(0);
    fetch(t0 + 1);
    barrier_lds();
    for (unsigned t = t0; t < t1; t++) {
        const int slot = (int)((t - t0) & 1u);
        if (!late) { // waves 0-3: multiply, then stage the next tile and store this one
            compute(slot);
            __builtin_amdgcn_sched_barrier(0);
            commit(slot ^ 1); // tile t + 1 (beyond the run: zeros / unused rows) into the slot tile t - 1 was read from
            fetch(t + 2 < t1 ? t + 2 : g.ntiles);
            epilogue(t);
        } else { // waves 4-7: finish the previous tile, stage, then multiply -- the other half of their SIMD mates' interval
            if (t > t0) epilogue(t - 1);
            commit(slot ^ 1);
            fetch(t + 2 < t1 ? t + 2 : g.ntiles);
            __builtin_amdgcn_sched_barrier(0);
            compute(slot);
        }
        barrier_lds();
    }
    if (late) epilogue(t1 - 1);
}

// ---------------------------------------------------------------------------------------------------------------------
static int stem_geom(const mhip_conv_f32_t *p, stem_geom_t *g, int frames) {
    memset(g, 0, sizeof(*g));
    if (p->stride_h != 2 || p->stride_w != 2 || p->pad_top != p->pad_left || (p->pad_top & 1) || p->pad_top < 0 || p->pad_top > 6) return 0;
    if (p->in_c < 1 || p->in_c > 4 || p->out_c < 1 || p->out_c > 32 || (p->kw & 1) || p->kw > 8 || p->kh < 1 || p->kh > 8) return 0;
    if (p->kh * p->kw / 2 > ST_NS * 4 || p->kh * p->kw < 8) return 0;
    if ((p->in_w & 3) || p->out_h % ST_TH || p->out_w % ST_TW) return 0;
    if ((p->out_h - 1) * 2 + p->kh - p->pad_top > p->in_h + 6 || (p->out_w - 1) * 2 + p->kw - p->pad_left > p->in_w + 6) return 0;
    g->in_c = p->in_c; g->kh = p->kh; g->kw = p->kw; g->pad = p->pad_top;
    g->H_in = p->in_h; g->W_in = p->in_w; g->H_out = p->out_h; g->W_out = p->out_w;
    const int dx = (4 - (p->pad_left & 3)) & 3; // columns between the aligned patch start and tap column 0 of tile column 0 (even)
    g->dxp = dx / 2;
    const int cols = (dx + 2 * ST_TW + p->kw - 2 + 3) & ~3;
    g->NG = cols / 4; g->PPW = cols / 2;
    g->PR = 2 * ST_TH + p->kh - 2;
    g->nunits = p->kh * p->kw / 2;
    g->nitems = g->PR * g->NG;
    if (g->nitems > 2 * ST_NT) return 0;
    g->slot_bytes = g->PR * g->PPW * 32;
    g->lds_bytes = 256 + 2 * g->slot_bytes;
    if (g->lds_bytes > 160 * 1024) return 0;
    g->tiles_x = p->out_w / ST_TW; g->tiles_y = p->out_h / ST_TH;
    const long nt = (long)frames * g->tiles_x * g->tiles_y;
    const size_t in_bytes = (size_t)(frames - 1) * p->in_stride + (size_t)p->in_c * p->in_h * p->in_w * 4;
    if (nt > 0x7ffffff0L || in_bytes > 0xfffffff0ull) return 0;
    g->ntiles = (unsigned)nt; g->in_bytes = (unsigned)in_bytes;
    return 1;
}

static uint16_t sbf16_rn(float x) {
    uint32_t b;
    memcpy(&b, &x, 4);
    if ((b & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((b >> 16) | 0x40u);
    return (uint16_t)((b + 0x7fffu + ((b >> 16) & 1u)) >> 16);
}
static float sbf16_val(uint16_t h) {
    const uint32_t b = (uint32_t)h << 16;
    float f;
    memcpy(&f, &b, 4);
    return f;
}
static void stem_shape(mhip_conv_f32_t *p, int out_c, int in_c, int kh, int kw, int stride, int pad, int in_h, int in_w, int out_h, int out_w) {
    memset(p, 0, sizeof(*p));
    p->out_c = out_c; p->in_c = in_c; p->kh = kh; p->kw = kw; p->stride_h = p->stride_w = stride; p->pad_top = p->pad_left = pad;
    p->in_h = in_h; p->in_w = in_w; p->out_h = out_h; p->out_w = out_w;
    p->in_stride = (size_t)in_c * in_h * in_w * 4; p->out_stride = (size_t)out_c * out_h * out_w * 4; p->frames = 1;
}

// Bytes of, and (w, out != NULL) the content of, the weight image conv_f32_stem loads into registers: two planes (hi, mid) of bf16
// [32 channels][20 units][8]: element par * 4 + ch of unit (ky, j) = w[oc][ch][ky][2 j + par] (channel slots >= in_c, units >=
// kh * kw / 2, channels >= out_c: zeros).  0 = not a shape this kernel takes.
extern "C" size_t mhip_conv_f32_stem_pack(int out_c, int in_c, int kh, int kw, int stride, int pad, int in_h, int in_w, int out_h, int out_w,
                                          const float *w, void *out) {
    mhip_conv_f32_t p;
    stem_shape(&p, out_c, in_c, kh, kw, stride, pad, in_h, in_w, out_h, out_w);
    stem_geom_t g;
    if (!stem_geom(&p, &g, 1)) return 0;
    const size_t rowe = (size_t)ST_NS * 4 * 8, planee = 32 * rowe, bytes = 2 * planee * 2;
    if (!w || !out) return bytes;
    memset(out, 0, bytes);
    uint16_t *hi = (uint16_t *)out, *mid = hi + planee;
    for (int oc = 0; oc < out_c; oc++)
        for (int ky = 0; ky < kh; ky++)
            for (int kx = 0; kx < kw; kx++)
                for (int ch = 0; ch < in_c; ch++) {
                    const float x = w[((size_t)(oc * (size_t)in_c + ch) * kh + ky) * kw + kx];
                    const uint16_t h = sbf16_rn(x);
                    const float hv = sbf16_val(h);
                    const size_t k = (size_t)oc * rowe + (size_t)(ky * (kw / 2) + kx / 2) * 8 + (size_t)(kx & 1) * 4 + ch;
                    hi[k] = h;
                    mid[k] = (hv - hv == 0.0f) ? sbf16_rn(x - hv) : 0;
                }
    return bytes;
}

static unsigned long g_stem_launches = 0;
extern "C" unsigned long mhip_conv_f32_stem_launches(void) { return g_stem_launches; }

// -2: not a shape this kernel takes, else the launch result
int conv_f32_try_stem(const mhip_conv_f32_t *p) {
    if (!p->w_patch || p->use_mfma != 3 || p->in_rec) return -2;
    stem_geom_t g;
    if (!stem_geom(p, &g, p->frames)) return -2;
    if (p->add && p->add_stride != p->out_stride) return -2;
    if (p->out_rec && (p->add || (p->out_c & 7))) return -2;
    static int cus = 0;
    if (!cus) {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipFuncSetAttribute((const void *)conv_f32_stem<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void *)conv_f32_stem<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return mhip_check(hipErrorUnknown, "conv_f32_stem attribute");
        cus = prop.multiProcessorCount;
    }
    int slots = 0;
    mhip_conv_i8_tune_get("persist_slots", &slots);
    unsigned gx = (unsigned)(slots > 0 ? slots : cus);
    if (gx > g.ntiles) gx = g.ntiles;
    g.per = (g.ntiles + gx - 1) / gx;
    gx = (g.ntiles + g.per - 1) / g.per;
    if (p->out_rec) hipLaunchKernelGGL(conv_f32_stem<true>, dim3(gx), dim3(ST_NT), (size_t)g.lds_bytes, mhip_stream_native(), *p, g, (const int8_t *)p->w_patch);
    else hipLaunchKernelGGL(conv_f32_stem<false>, dim3(gx), dim3(ST_NT), (size_t)g.lds_bytes, mhip_stream_native(), *p, g, (const int8_t *)p->w_patch);
    g_stem_launches++;
    return mhip_check(hipGetLastError(), "conv_f32_stem");
}
