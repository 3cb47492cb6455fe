// conv_f32.hip -- float32 convolution, NCHW / OIHW, for gfx950.
//
// Replaces reference src/mars/mxu_conv.c:673-710 (conv2d_float32_mxu; scalar even on the camera).  The reference
// accumulates sequentially
//     sum = bias; for ic, kh, kw (in-image taps only): sum += in * w
// with one rounded multiply and one rounded add per tap.  Two forms:
//
//  conv_f32_kernel (mode 0, "exact"): that order and those roundings per output element (no FMA contraction: built with
//      -ffp-contract=off), so it is BIT-IDENTICAL to the reference.  One lane per output element; f32 VALU bound.
//  conv_f32_mfma   (mode 1, "mfma"): implicit GEMM on v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: an exact-product
//      fused multiply-add chain, 157 TF peak).  D[oc][pixel] = bias[oc] + sum_k W[oc][k] * X[k][pixel], k = (ic, ky, kx) in
//      the reference's order, K consumed four taps per instruction.  Differs from the reference only by the fused
//      rounding (one rounding per tap instead of two): inside north_star's 1e-4 tolerance, not bit-equal.  Padded taps
//      multiply a zero instead of being skipped (equal up to the sign of a zero sum).
//
// Which form a layer gets is the host's decision (mars_model.c, plan_conv / f32_policy): a byte-wise MAXPOOL over float
// bytes (reference mars_runtime.c:919-957 runs int8 byte logic whatever the dtype) is discontinuous in its input, so by
// default every convolution UPSTREAM of such a pool stays exact and only the rest takes the matrix cores;
// mars_hip_set_tuning("f32_mfma", 2) uses them everywhere (tolerance verified on the config-5 output), 0 nowhere; 3 / 4 (round 4) =
// everywhere AND on the bf16 matrix cores with every operand split into two / three bf16 pieces, three / six piece products per
// product (conv_f32_split.hip); 3 is the config-5 benchmark.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../expf_exact.h"
#include "../mhip.h"

extern "C" hipStream_t mhip_stream_native(void);
extern "C" int mhip_check(hipError_t e, const char *what);

typedef float v4f __attribute__((ext_vector_type(4)));

// fused SiLU as the exporter writes it (conv -> SIGMOID -> MUL, reference mars_runtime.c:742-749 and :807-816 float forms):
// s = 1.0f / (1.0f + expf(-v)); out = v * s -- every step rounded as the reference rounds it, expf as this image's libm
// (expf_exact.h), so folding the two element-wise layers into the convolution's epilogue changes no bit
__device__ __forceinline__ float silu_f32(float v) {
    const float s = 1.0f / (1.0f + expf_exact(-v, expf_exact_tab));
    return v * s;
}

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
    const size_t oi = ((size_t)oc * p.out_h + oy) * p.out_w + ox;
    const float r = p.silu ? silu_f32(acc) : acc;
    out[oi] = p.add ? r + ((const float *)((const char *)p.add + (size_t)f * p.add_stride))[oi] : r;
}

// ---------------------------------------------------------------------------------
// implicit GEMM on the f32 matrix cores.  Workgroup = 256 threads = 4 waves (2 along oc x 2 along pixels); tile = 64
// output channels x 128 pixels (pixels of all frames flattened, frame-major); every wave owns 32 x 64 = 2 x 4 MFMA
// tiles (32 accumulator registers).  K step = 16 taps: the weight tile [64][16] is one 16-byte load per thread (K is
// contiguous in OIHW rows), the input tile [16][128] is gathered element-wise (8 per thread: in NCHW the 8 consecutive
// pixels of one tap are contiguous floats unless a row / frame / border intervenes -- the per-element index arithmetic
// hides behind the 32-cycle MFMAs).  Both tiles are double buffered in LDS; row pitches are 16 mod 32 floats so that
// the fragment reads (16 lanes x 4 k-rows per ds_read_b32 group) are conflict free.
#define F_BM 64
#define F_BN 128
#define F_BK 16
#define F_APITCH 80   // floats per k-row of the weight tile  ([k][m], m contiguous): 64 + 16
#define F_BPITCH 144  // floats per k-row of the input tile   ([k][n], n contiguous): 128 + 16
struct fdiv_t {
    unsigned m, s1, s2;
};
__device__ __forceinline__ unsigned fdivf(unsigned n, const fdiv_t d) {
    const unsigned q = __umulhi(d.m, n);
    return (q + ((n - q) >> d.s1)) >> d.s2;
}
static fdiv_t make_fdiv(unsigned d) {
    fdiv_t r;
    unsigned l = 0;
    while ((1ull << l) < d) l++;
    r.m = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    r.s1 = l < 1 ? l : 1;
    r.s2 = l > 0 ? l - 1 : 0;
    return r;
}

__global__ __launch_bounds__(256) void conv_f32_mfma(const mhip_conv_f32_t p, const unsigned total_pix, const int K, const int nks,
                                                     const unsigned npt, const fdiv_t dhw, const fdiv_t dow, const fdiv_t dtaps,
                                                     const fdiv_t dkw) {
    __shared__ __attribute__((aligned(16))) float As[2][F_BK * F_APITCH];
    __shared__ __attribute__((aligned(16))) float Bs[2][F_BK * F_BPITCH];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv & 1, wn = wv >> 1; // wave tile: channels [wm*32, +32), pixels [wn*64, +64)
    const unsigned pt = blockIdx.x % npt, ot = blockIdx.x / npt;
    const unsigned p0 = pt * F_BN;
    const int oc0 = (int)ot * F_BM;
    const unsigned hw = (unsigned)(p.out_h * p.out_w);
    const int taps = p.kh * p.kw;
    const size_t plane = (size_t)p.in_h * p.in_w;

    // ---- this thread's share of the tile loads
    // weights: row oc0 + tid/4, K floats (tid%4)*4 .. +3 of the step
    const int arow = tid >> 2, akc = (tid & 3) * 4;
    const bool arow_ok = oc0 + arow < p.out_c;
    const float *wrow = p.w + (size_t)(arow_ok ? oc0 + arow : 0) * K;
    // input: tap row kb = tid/16 of the step, pixels p0 + (tid%16)*8 .. +7
    const int kb = tid >> 4, nb = (tid & 15) * 8;
    int iy0[8], ix0[8];
    const float *fbase[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const unsigned px = p0 + (unsigned)(nb + e);
        const bool valid = px < total_pix;
        const unsigned f = fdivf(valid ? px : 0u, dhw), rem = (valid ? px : 0u) - f * hw;
        const int oy = (int)fdivf(rem, dow), ox = (int)rem - oy * p.out_w;
        iy0[e] = valid ? oy * p.stride_h - p.pad_top : -(1 << 28); // invalid pixels fail every bounds test
        ix0[e] = ox * p.stride_w - p.pad_left;
        fbase[e] = (const float *)((const char *)p.in + (size_t)f * p.in_stride);
    }
    v4f areg;
    float breg[8];
    auto fetch = [&](int ks) {
        const int ka = ks * F_BK + akc;
        areg = (v4f){0.f, 0.f, 0.f, 0.f};
        if (arow_ok) {
            if (ka + 3 < K && ((K & 3) == 0)) areg = *(const v4f *)(wrow + ka); // 16-byte aligned when K % 4 == 0
            else {
#pragma unroll
                for (int j = 0; j < 4; j++) areg[j] = ka + j < K ? wrow[ka + j] : 0.f;
            }
        }
        const int k = ks * F_BK + kb;
        const bool kok = k < K;
        const unsigned ic = fdivf((unsigned)(kok ? k : 0), dtaps), r = (unsigned)(kok ? k : 0) - ic * (unsigned)taps;
        const int ky = (int)fdivf(r, dkw), kx = (int)r - ky * p.kw;
        const size_t coff = (size_t)ic * plane;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int iy = iy0[e] + ky, ix = ix0[e] + kx;
            const bool ok = kok && (unsigned)iy < (unsigned)p.in_h && (unsigned)ix < (unsigned)p.in_w;
            breg[e] = ok ? fbase[e][coff + (size_t)iy * p.in_w + ix] : 0.f;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; j++) As[buf][(akc + j) * F_APITCH + arow] = areg[j];
        *(v4f *)&Bs[buf][kb * F_BPITCH + nb] = (v4f){breg[0], breg[1], breg[2], breg[3]};
        *(v4f *)&Bs[buf][kb * F_BPITCH + nb + 4] = (v4f){breg[4], breg[5], breg[6], breg[7]};
    };

    // accumulators start at the bias: lane holds rows (channels) 4*(lane/16) + j of an MFMA tile
    v4f acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; a++) {
        v4f b = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int oc = oc0 + wm * 32 + a * 16 + (lane >> 4) * 4 + j;
                b[j] = oc < p.out_c ? p.bias[oc] : 0.f;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) acc[a][c] = b;
    }

    fetch(0);
    commit(0);
    __syncthreads();
    const int fm = lane & 15, fk = lane >> 4; // fragment element of this lane: row / column fm, k-row fk of a 4-tap group
    for (int ks = 0; ks < nks; ks++) {
        const int buf = ks & 1;
        if (ks + 1 < nks) fetch(ks + 1); // global loads in flight behind this step's MFMAs
        const float *as = As[buf] + wm * 32 + fm, *bs = Bs[buf] + wn * 64 + fm;
#pragma unroll
        for (int g = 0; g < F_BK / 4; g++) {
            float af[2], bf[4];
#pragma unroll
            for (int a = 0; a < 2; a++) af[a] = as[(g * 4 + fk) * F_APITCH + a * 16];
#pragma unroll
            for (int c = 0; c < 4; c++) bf[c] = bs[(g * 4 + fk) * F_BPITCH + c * 16];
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int c = 0; c < 4; c++) acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bf[c], acc[a][c], 0, 0, 0);
        }
        if (ks + 1 < nks) commit(buf ^ 1); // the other buffer: last read in step ks - 1, every wave is past that step's barrier
        __syncthreads();
    }
    // store: lane holds channels 4*(lane/16)+j, pixel fm of each tile; 16 lanes write 16 consecutive floats of a channel row
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const unsigned px = p0 + (unsigned)(wn * 64 + c * 16 + fm);
        if (px >= total_pix) continue;
        const unsigned f = fdivf(px, dhw), rem = px - f * hw;
        float *out = (float *)((char *)p.out + (size_t)f * p.out_stride);
        const float *addp = p.add ? (const float *)((const char *)p.add + (size_t)f * p.add_stride) : nullptr;
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int oc = oc0 + wm * 32 + a * 16 + (lane >> 4) * 4 + j;
                if (oc < p.out_c) {
                    const float r = p.silu ? silu_f32(acc[a][c][j]) : acc[a][c][j];
                    out[(size_t)oc * hw + rem] = addp ? r + addp[(size_t)oc * hw + rem] : r;
                }
            }
    }
}

int conv_f32_try_split(const mhip_conv_f32_t *p); // conv_f32_split.hip: -2 = not a shape it takes
int conv_f32_try_patch(const mhip_conv_f32_t *p); // conv_f32_patch.hip: -2 = not a shape it takes (or no image packed)
int conv_f32_try_stem(const mhip_conv_f32_t *p);  // conv_f32_stem.hip: likewise

static int g_f32_mode = -1; // -1: environment not read yet
extern "C" int mhip_conv_f32_mode(int set) { // set >= 0: new mode; returns the mode in force
    if (g_f32_mode < 0) {
        const char *e = getenv("MARS_HIP_F32_MFMA");
        g_f32_mode = e ? atoi(e) : 1;
        if (g_f32_mode < 0 || g_f32_mode > 4) g_f32_mode = 1;
    }
    if (set >= 0 && set <= 4) g_f32_mode = set;
    return g_f32_mode;
}

extern "C" int mhip_conv_f32(const mhip_conv_f32_t *p) {
    if (!p || !p->in || !p->out || !p->w) return -1;
    if (p->frames <= 0 || p->in_h <= 0 || p->in_w <= 0 || p->in_c <= 0 || p->out_h <= 0 || p->out_w <= 0 ||
        p->out_c <= 0 || p->kh <= 0 || p->kw <= 0 || p->stride_h <= 0 || p->stride_w <= 0)
        return -1;
    if (p->out_c > 65535 || p->frames > 65535) return -1;
    const long hw = (long)p->out_h * p->out_w, total = hw * p->frames, K = (long)p->in_c * p->kh * p->kw;
    if (p->use_mfma >= 2) { // the bf16 matrix cores on split operands (conv_f32_split.hip): 2 = six piece products, 3 = three
        int rc = p->k_limit_required ? -2 : conv_f32_try_stem(p); // the 3-channel first layer, three piece products (round 5)
        if (rc != -2) return rc;
        rc = p->k_limit_required ? -2 : conv_f32_try_patch(p); // k x k layers, three piece products: the patch-staged form (round 5)
        if (rc != -2) return rc;
        rc = p->w_split ? conv_f32_try_split(p) : -2;
        if (rc != -2) return rc;
    }
    // a shifted view of another tensor (virtual_concat_f32): what lies behind k_limit planes is not zero -- only conv_f32_split stops there
    if (p->k_limit_required) return mhip_check(hipErrorInvalidValue, "conv_f32: a K-limited input view and no kernel that honours the limit");
    // record-format tensors exist between two of the kernels above only (the planner's pairing: mars_plan.c); nothing below reads or writes them
    if (p->in_rec || p->out_rec) return mhip_check(hipErrorInvalidValue, "conv_f32: record-format operand and no kernel for it");
    if (p->use_mfma && total <= 0x7fffffffL - F_BN && K <= 0x7fffffffL - F_BK) {
        const unsigned npt = (unsigned)((total + F_BN - 1) / F_BN), noc = (unsigned)((p->out_c + F_BM - 1) / F_BM);
        if ((unsigned long long)npt * noc <= 0x7fffffffull) {
            hipLaunchKernelGGL(conv_f32_mfma, dim3(npt * noc), dim3(256), 0, mhip_stream_native(), *p, (unsigned)total, (int)K,
                               (int)((K + F_BK - 1) / F_BK), npt, make_fdiv((unsigned)hw), make_fdiv((unsigned)p->out_w),
                               make_fdiv((unsigned)(p->kh * p->kw)), make_fdiv((unsigned)p->kw));
            return mhip_check(hipGetLastError(), "conv_f32_mfma");
        }
    }
    dim3 grid((unsigned)((hw + 255) / 256), (unsigned)p->out_c, (unsigned)p->frames);
    hipLaunchKernelGGL(conv_f32_kernel, grid, dim3(256), 0, mhip_stream_native(), *p);
    return mhip_check(hipGetLastError(), "conv_f32");
}
