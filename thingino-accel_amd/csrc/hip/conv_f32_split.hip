// conv_f32_split.hip -- float32 convolution (NCHW / OIHW, reference src/mars/mxu_conv.c:673-710) on the bf16 matrix cores
// by operand splitting.  Round 4, verdict item 3; mhip_conv_f32_t.use_mfma == 3 / 2 = tuning "f32_mfma" 3 / 4.
//
// The f32-input MFMA (conv_f32.hip, v_mfma_f32_16x16x4_f32) runs at the f32 VECTOR rate, 1/16 of the bf16 matrix rate, and
// gfx950 has no xf32: config 5 sat at 0.37 of a 157 TFLOP/s ceiling.  Here every float is cut, exactly, into bf16 pieces by
// round-to-nearest,  hi = bf16(x), mid = bf16(x - hi), lo = x - hi - mid  (every subtraction exact: x - hi has at most 16
// significant bits, x - hi - mid at most 8), and a product a * b is summed from piece products, each exact in f32, added into
// f32 accumulators by v_mfma_f32_16x16x32_bf16:
//   NPL = 2 (use_mfma 3): two pieces, THREE products  a.hi b.hi + a.hi b.mid + a.mid b.hi   -- relative error per product <= 2^-16,
//                         either sign; 8e-7 worst relative error on the config-5 twin's outputs; the benchmark's mode;
//   NPL = 3 (use_mfma 2): three pieces, SIX products  ... + a.mid b.mid + a.hi b.lo + a.lo b.hi -- what is dropped is below 2^-23
//                         relative, the size of one f32 rounding.
// Inside north_star's 1e-4 tolerance, not bit-equal (like conv_f32_mfma; the exact-order kernel stays where a byte-wise
// consumer follows).  Peak for the roofline: 2.5 PFLOP/s / 3 (or / 6) of float32 work.
//
// Implicit GEMM  D[oc][pixel] = bias[oc] + sum_k W[oc][k] X[k][pixel],  k = (ic, ky, kx) in the reference's order.  Workgroup =
// 512 threads = 8 waves; tile = BM output channels x 256 pixels (all frames flattened), K step = 32 taps.  PERSISTENT over
// pixel tiles (grid = what the device holds x channel tiles): the K pipeline runs through the tile boundary, so a tile's
// stores overlap the next tile's first loads.
//  * weights: cut into their planes ONCE, on the host at load time (mhip_conv_f32_split_pack: [plane][oc_pad][k_pad] bf16);
//    a step's tile is one 16-byte copy per thread and plane;
//  * input: gathered through registers with 16-BYTE loads, split there and written as K-contiguous bf16 rows -- the MFMA's
//    fragment layout -- into the planes of the LDS tile.  Stride 1 (GATHER 1): a lane owns 4 consecutive pixels of a map row
//    and loads them for one tap in one instruction; stride 2 (GATHER 2): a lane owns 2 pixels and a load brings taps (kx,
//    kx + 1) of both -- kernel rows are padded to an even length there (one zero weight column).  A tap left of the image
//    (pad 1) loads one element further right and shifts; columns outside the image are masked by a per-lane bit table, rows
//    outside by an out-of-range offset the buffer unit turns into zeros.  GATHER 0 (any other geometry): one dword per tap.
//    A fetch ONLY issues loads (a scheduler fence follows it; left alone the compiler sinks the loads to the end of the step);
//    shifts and masks are applied by the commit one step later from a per-lane `meta` word;
//  * LDS: A rows as in the int8 kernels (64-byte rows, chunk swizzle).  B rows (pixels) are dealt over four 64-row blocks
//    (pixel r -> block r % 4, row r / 4) with the 16-byte chunk XOR-ed by (block ^ row / 4 % 4), laid out so that the
//    4-pixel-per-lane writes and the 16-pixel fragment reads spread over the banks (SQ counters: 22 % of the LDS cycles are
//    still conflicts -- LDS is busy a fifth of the time, not the bound);
//  * two stages, two register sets: step ks + 2's loads are in flight while step ks multiplies and step ks + 1's operands
//    are split and written (sched_group_barrier interleaves them with the MFMAs: two vector instructions fit in an MFMA's shadow).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../expf_exact.h"
#include "../mhip.h"

extern "C" hipStream_t mhip_stream_native(void);
extern "C" int mhip_check(hipError_t e, const char *what);

typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define S_BN 256  // pixels per workgroup
#define S_BK 32   // taps per K step = one v_mfma_f32_16x16x32_bf16
#define S_NT 512
#ifdef SPLIT_STAMPS // diagnostic build (tools/stamps_build.sh splitstamps; read with tools/split_stamps.py): where a K step goes
__device__ unsigned long long split_stamp_sums[8];
extern "C" int mhip_split_stamps(unsigned long long *out, int reset) {
    unsigned long long z[8] = {0};
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(split_stamp_sums), sizeof z) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(split_stamp_sums), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
#define STAMP(i)                                                                  \
    do {                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                        \
        const unsigned long long t_ = __builtin_readcyclecounter();              \
        st_acc[i] += t_ - st_last;                                                \
        st_last = t_;                                                             \
        __builtin_amdgcn_sched_barrier(0);                                        \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
#ifndef SPLIT_ABL // timing-only ablations (tools/stamps_build.sh split N; wrong results): 1 no MFMAs, 2 no gather loads, 4 no
#define SPLIT_ABL 0 // split / LDS writes of the input, 8 no fragment reads, 16 plain stores (no SiLU), 32 no stores, 64 aligned gathers
#endif

__device__ __forceinline__ float silu_split(float v) { // as conv_f32.hip: the exporter's SiLU with the reference's roundings
    const float s = 1.0f / (1.0f + expf_exact(-v, expf_exact_tab));
    return v * s;
}
// three piece products (relative error per product up to 2^-16) do not need libm's last bit: v_exp_f32 and v_rcp_f32, 1 ulp each --
// the exact form above is double-precision arithmetic and a table, a fifth of the kernel's time on every layer
__device__ __forceinline__ float silu_fast(float v) {
    const float e = __builtin_amdgcn_exp2f(v * -1.44269504088896341f); // exp(-v); +inf for v << 0: rcp -> 0, v * 0 = -0
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ int a_lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 2)) << 4); }
__device__ __forceinline__ int b_lds_off(int r, int chunk) { // pixel r of the tile, 16-byte chunk (8 taps) of its 64-byte row
    return ((((r & 3) << 6) + (r >> 2)) << 6) + (((chunk ^ r ^ (r >> 4)) & 3) << 4);
}

struct sdiv_t {
    unsigned m, s1, s2;
};
__device__ __forceinline__ unsigned sdiv(unsigned n, const sdiv_t d) {
    const unsigned q = __umulhi(d.m, n);
    return (q + ((n - q) >> d.s1)) >> d.s2;
}
static sdiv_t make_sdiv(unsigned d) {
    sdiv_t r;
    unsigned l = 0;
    while ((1ull << l) < d) l++;
    r.m = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    r.s1 = l < 1 ? l : 1;
    r.s2 = l > 0 ? l - 1 : 0;
    return r;
}

// N floats -> N bf16 in each of NPL planes (N / 2 dwords each), round-to-nearest split (v_cvt_pk_bf16_f32 converts and packs
// two at a time): hi = bf16(x), mid = bf16(x - hi), lo = x - hi - mid.  Every subtraction is exact (x - hi has at most 16
// significant bits, x - hi - mid at most 8, a bf16 value), so x == hi + mid + lo exactly; with two planes |x - hi - mid| <=
// 2^-17 |x|, errors of either sign.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int N, int NPL>
__device__ __forceinline__ void splitn(const float (&x)[N], int (&hi)[N / 2], int (&mid)[N / 2], int (&lo)[N / 2]) {
#pragma unroll
    for (int i = 0; i < N / 2; i++) { // (element 2i, element 2i + 1) -> one dword per plane, element 2i in the low half
        const f32x2 v = {x[2 * i], x[2 * i + 1]};
        const int h = __builtin_bit_cast(int, __builtin_convertvector(v, bf16x2));
        const f32x2 hf = {__int_as_float(h << 16), __int_as_float(h & (int)0xffff0000)};
        const f32x2 r = v - hf;
        const int m = __builtin_bit_cast(int, __builtin_convertvector(r, bf16x2));
        hi[i] = h;
        mid[i] = m;
        if (NPL == 3) {
            const f32x2 mf = {__int_as_float(m << 16), __int_as_float(m & (int)0xffff0000)};
            lo[i] = __builtin_bit_cast(int, __builtin_convertvector(r - mf, bf16x2)); // exact
        } else lo[i] = 0;
    }
}

struct split_args_t {
    unsigned total_pix, npt, per, in_bytes; // pixels, pixel tiles, tiles per workgroup
    int K;       // taps in the packed K space (kernel rows padded to kwp)
    int kp;      // row length of the weight planes (bf16 elements): K rounded up to 64, + 64 of slack (the loop fetches two steps ahead)
    int nks;     // K steps: roundup64(K) / 32 (even)
    int kwp;     // kernel row length in the packed K space (kw, or kw + 1 for GATHER 2 with an odd kw)
    int oc_pad;  // rows of a weight plane
    sdiv_t dhw, dow, dtaps, dkwp;
    // a PAIR of convolutions over the same input in one grid (mhip_conv_f32_pair: C3's cv1 + cv2 -- same geometry, same out_c): channel tiles
    // blockIdx.y >= noc1 belong to the second one (its weight planes, bias and output); the two workgroups of a pixel run walk the same
    // tiles side by side, the second one's input reads hit L2.  noc1 == 0: one convolution
    unsigned noc1;
    const void *w_split2;
    const float *bias2;
    float *out2;
};

// BM = output channels per workgroup (128 | 64 | 32); waves: WM along channels x WN along pixels, WM * WN == 8
// RECOUT (mhip_conv_f32_t.out_rec, two pieces only): the results leave as RECORDS -- [out_c / 8][H][W] x 32 bytes = [8 x bf16 hi | 8 x bf16 mid] of
// 8 consecutive channels of one pixel -- for the one k x k convolution that reads them (conv_f32_prec in conv_f32_patch.hip: its staging
// phase is then plain LDS-DMA).  The MFMA operands change places (D = W X^T), so a lane ends with 4 consecutive CHANNELS of one pixel.
template <int BM, int WM, int WN, int GATHER, int NPL, bool RECOUT = false>
__global__ __launch_bounds__(S_NT, BM == 32 && NPL == 2 && !RECOUT ? 4 : 2) void conv_f32_split(const mhip_conv_f32_t p_, const split_args_t g) {
    mhip_conv_f32_t p = p_;
    unsigned by = blockIdx.y;
    // fz: the pair shares ONE channel tile (BM == 2 * out_c: rows 0 .. BM / 2 - 1 are the first convolution's channels, the rest the
    // second's) -- one workgroup reads the input tile once for both (the 160 x 160 / 80 x 80 C3s of the twins, where these layers
    // are bound by exactly those bytes); otherwise channel tiles blockIdx.y >= noc1 belong to the second one
    const bool fz = g.noc1 == 0xffffffffu;
    if (!fz && g.noc1 && by >= g.noc1) {
        by -= g.noc1;
        p.w_split = g.w_split2; p.bias = g.bias2; p.out = g.out2;
    }
    const int out_c_eff = fz ? BM : p.out_c;
    constexpr int TM = BM / WM, TN = S_BN / WN; // wave tile
    constexpr int MI = TM / 16, NI = TN / 16;   // MFMA tiles per wave
    constexpr int APLANE = BM * 64, BPLANE = S_BN * 64;
    constexpr int NPROD = NPL == 3 ? 6 : 3; // piece products per product
    constexpr int STAGE = NPL * (APLANE + BPLANE);
    constexpr int AE = BM * S_BK / S_NT;        // weight elements per thread, plane and step: 8 | 4 | 2 consecutive taps of one row
    constexpr int ATPR = S_BK / AE;             // threads per weight row
    constexpr int AD = AE / 2;                  // ... in dwords
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv % WM, wn = wv / WM;
    // persistent over pixel tiles: this workgroup keeps its channel tile (blockIdx.y) and walks pixel tiles blockIdx.x,
    // + gridDim.x, ...  The K pipeline runs THROUGH the tile boundary (the last two steps of a tile fetch steps 0 and 1 of the
    // next), so a tile's stores and the next tile's first loads overlap -- layers with few K steps (the stem: 4) otherwise
    // pay two exposed memory latencies and a store drain per 256 pixels
    const int oc0 = (int)by * BM;
    const unsigned hw = (unsigned)(p.out_h * p.out_w);
    const int K = g.K, kwp = g.kwp;
    const int taps = p.kh * kwp;
    const int plane = p.in_h * p.in_w;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, (int)g.in_bytes, 0x00020000);

    // ---- this thread's share of the input gather: PPL consecutive pixels of one map row, TPL taps per step
    constexpr int PPL = GATHER == 1 ? 4 : (GATHER == 2 ? 2 : 1); // pixels per lane
    constexpr int TPL = 16 / PPL;                                 // taps per lane and step
    const int pl_ = tid % (S_BN / PPL);                           // pixel group inside the tile
    const int tg = __builtin_amdgcn_readfirstlane(tid / (S_BN / PPL)); // tap group (wave-uniform): taps tg * TPL .. of the step
    unsigned vbase;     // byte offset of the window origin (channel 0) of the lane's FIRST pixel; "negative" values wrap
    unsigned rowbits, colbits; // kernel rows inside the image; (tap column, element) pairs inside the image
    int ix0;
    auto setup = [&](unsigned pt) __attribute__((always_inline)) { // the fetch side's view of pixel tile pt (beyond the last: nothing valid)
        const unsigned px = pt * S_BN + (unsigned)(pl_ * PPL);
        const bool valid = pt < g.npt && px < g.total_pix; // (GATHER 1 / 2: out_w % PPL == 0, so the lane's pixels share row, frame and validity)
        const unsigned f = sdiv(valid ? px : 0u, g.dhw), rem = (valid ? px : 0u) - f * hw;
        const int oy = (int)sdiv(rem, g.dow), ox = (int)rem - oy * p.out_w;
        const int iy0 = oy * p.stride_h - p.pad_top;
        ix0 = ox * p.stride_w - p.pad_left;
        vbase = (unsigned)(f * (unsigned)p.in_stride) + (unsigned)((iy0 * p.in_w + ix0) * 4);
        rowbits = 0; colbits = 0;
        if (valid)
            for (int ky = 0; ky < p.kh; ky++)
                if ((unsigned)(iy0 + ky) < (unsigned)p.in_h) rowbits |= 1u << ky;
        if (GATHER == 0) {
            for (int kx = 0; kx < kwp; kx++)
                if (kx < p.kw && (unsigned)(ix0 + kx) < (unsigned)p.in_w) colbits |= 1u << kx;
        } else if (GATHER == 1) { // element i of the load for tap column kx = pixel i: column ix0 + i + kx
            for (int kx = 0; kx < p.kw; kx++)
                for (int i = 0; i < 4; i++)
                    if ((unsigned)(ix0 + i + kx) < (unsigned)p.in_w) colbits |= 1u << (kx * 4 + i);
        } else { // element i of the load for tap columns (kx, kx + 1), kx even: pixel i / 2, tap kx + i % 2: column ix0 + 2 (i / 2) + kx + i % 2
            for (int kx = 0; kx < kwp; kx += 2)
                for (int i = 0; i < 4; i++)
                    if (kx + (i & 1) < p.kw && (unsigned)(ix0 + 2 * (i >> 1) + kx + (i & 1)) < (unsigned)p.in_w) colbits |= 1u << ((kx >> 1) * 4 + i);
        }
    };
    // this workgroup's run of pixel tiles: an even split of the tile list over the grid's x extent
    const unsigned pt_first = (unsigned)(((unsigned long long)blockIdx.x * g.npt) / gridDim.x);
    setup(pt_first);
    // weights: row oc0 + tid / ATPR of every plane, taps (tid % ATPR) * AE .. + AE - 1 of the step
    const int arow = tid / ATPR, akc = (tid % ATPR) * AE;
    const int8_t *wrow = fz && arow >= BM / 2 ? (const int8_t *)g.w_split2 + ((size_t)(arow - BM / 2) * g.kp + akc) * 2
                                              : (const int8_t *)p.w_split + ((size_t)(oc0 + arow) * g.kp + akc) * 2;
    const size_t wplane = (size_t)g.oc_pad * g.kp * 2;

    v4i bregs[2][4]; // GATHER 1 / 2: one 16-byte load each; GATHER 0: four dword loads each
    unsigned metas[2], smetas[2];
    int aregs[2][NPL][AD];
    // fetch ISSUES loads and nothing else (a scheduler fence follows it): what has to happen to the loaded vectors -- the shift
    // of a load that started one element late, the column masks -- is recorded per lane in `meta` (per load: 4 mask bits, 1
    // shift bit) and per wave in `smeta` (loads whose tap can start left of the image) and applied by commit one step later
    auto fetch = [&](int ks, v4i (&breg)[4], int (&areg)[NPL][AD], unsigned &meta, unsigned &smeta) __attribute__((always_inline)) {
        meta = 0; smeta = 0;
        // the lane's taps of this step: (ic, ky, kx) carried from the first one (wave-uniform: scalar registers)
        const int k0 = ks * S_BK + tg * TPL;
        int ic = (int)sdiv((unsigned)(k0 < K ? k0 : 0), g.dtaps);
        const int t0 = (k0 < K ? k0 : 0) - ic * taps;
        int ky = (int)sdiv((unsigned)t0, g.dkwp), kx = t0 - ky * kwp;
        int soff = (ic * plane + ky * p.in_w + kx) * 4; // byte offset of tap (ic, ky, kx) relative to the window origin
        constexpr int STEP = GATHER == 2 ? 2 : 1;       // taps per load
#pragma unroll
        for (int j = 0; j < TPL / STEP; j++) {
            const bool kok = k0 + j * STEP < K;
            const bool rowok = kok && ((rowbits >> ky) & 1u) != 0u;
            if (GATHER == 0) {
                const bool ok = rowok && ((colbits >> kx) & 1u) != 0u;
                // (the per-lane part alone may be "negative" -- first row / column of frame 0 -- and a vector offset beyond the
                // range reads zero whatever a scalar offset would add: the sum goes into the vector offset)
                if (SPLIT_ABL & 2) breg[j >> 2][j & 3] = (int)(vbase + (unsigned)soff) | (int)ok;
                else breg[j >> 2][j & 3] = (int)__builtin_amdgcn_raw_buffer_load_b32(xrs, ok ? vbase + (unsigned)soff : 0xffffffffu, 0, 0);
            } else {
                const bool lsh = ix0 + kx < 0; // the load would start left of the image (pad 1, first group of a row): start one
                                               // element later and shift -- the element shifted in is masked below anyway
                unsigned vo = rowok ? vbase + (unsigned)soff + (lsh ? 4u : 0u) : 0xffffffffu;
                if (SPLIT_ABL & 64) vo &= ~15u; // ablation: every gather 16-byte aligned (wrong data): what do the unaligned ones cost?
                if (SPLIT_ABL & 2) breg[j] = (v4i){(int)vo, (int)vo + 1, (int)vo + 2, (int)vo + 3};
                else breg[j] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, 0, 0));
                const unsigned bits = (colbits >> ((GATHER == 2 ? (kx >> 1) : kx) * 4)) & 15u;
                meta |= (bits | (lsh ? 16u : 0u)) << (5 * j);
                if (kx < p.pad_left) smeta |= 1u << j; // wave-uniform: only such taps can start left of the image
            }
            kx += STEP; soff += 4 * STEP;
            if (kx >= kwp) {
                kx = 0; ky++; soff += (p.in_w - kwp) * 4;
                if (ky == p.kh) { ky = 0; ic++; soff += (plane - p.kh * p.in_w) * 4; }
            }
        }
#pragma unroll
        for (int pl = 0; pl < NPL; pl++) {
            const int8_t *src = wrow + pl * wplane + (size_t)ks * (S_BK * 2);
            if (AE == 8) { const v4i t = *(const v4i *)src; areg[pl][0] = t[0]; areg[pl][1 % AD] = t[1]; areg[pl][2 % AD] = t[2]; areg[pl][3 % AD] = t[3]; }
            else if (AE == 4) { const int2 t = *(const int2 *)src; areg[pl][0] = t.x; areg[pl][1 % AD] = t.y; }
            else areg[pl][0] = *(const int *)src;
        }
    };
    auto commit = [&](int buf, const v4i (&braw)[4], const int (&areg)[NPL][AD], const unsigned meta, const unsigned smeta) __attribute__((always_inline)) {
        int8_t *st = lds + buf * STAGE;
        int8_t *bp = st + NPL * APLANE;
        float breg[16];
        if (GATHER == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) breg[i] = __int_as_float(braw[i >> 2][i & 3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int e[4] = {braw[j][0], braw[j][1], braw[j][2], braw[j][3]};
                if ((smeta >> j) & 1u) {
                    const bool lsh = ((meta >> (5 * j + 4)) & 1u) != 0u;
                    e[3] = lsh ? e[2] : e[3]; e[2] = lsh ? e[1] : e[2]; e[1] = lsh ? e[0] : e[1];
                }
#pragma unroll
                for (int i = 0; i < 4; i++) breg[4 * j + i] = __int_as_float(e[i] & __builtin_amdgcn_sbfe((int)meta, 5 * j + i, 1)); // 0 / ~0
            }
        }
        if (SPLIT_ABL & 4) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("" ::"v"(breg[i]));
        } else if (GATHER == 1) { // pixel i of the lane: taps tg * 4 .. + 3 = half a chunk
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float x[4] = {breg[i], breg[4 + i], breg[8 + i], breg[12 + i]};
                int hi[2], mid[2], lo[2];
                splitn<4, NPL>(x, hi, mid, lo);
                const int off = b_lds_off(pl_ * 4 + i, tg >> 1) + (tg & 1) * 8;
                *(int2 *)(bp + off) = make_int2(hi[0], hi[1]);
                *(int2 *)(bp + BPLANE + off) = make_int2(mid[0], mid[1]);
                if (NPL == 3) *(int2 *)(bp + 2 * BPLANE + off) = make_int2(lo[0], lo[1]);
            }
        } else if (GATHER == 2) { // pixel q of the lane: taps tg * 8 .. + 7 = one chunk; load j holds (q, tap 2j) at 2q, (q, 2j + 1) at 2q + 1
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const float x[8] = {breg[2 * q], breg[2 * q + 1], breg[4 + 2 * q], breg[5 + 2 * q], breg[8 + 2 * q], breg[9 + 2 * q], breg[12 + 2 * q], breg[13 + 2 * q]};
                int hi[4], mid[4], lo[4];
                splitn<8, NPL>(x, hi, mid, lo);
                const int off = b_lds_off(pl_ * 2 + q, tg);
                *(v4i *)(bp + off) = (v4i){hi[0], hi[1], hi[2], hi[3]};
                *(v4i *)(bp + BPLANE + off) = (v4i){mid[0], mid[1], mid[2], mid[3]};
                if (NPL == 3) *(v4i *)(bp + 2 * BPLANE + off) = (v4i){lo[0], lo[1], lo[2], lo[3]};
            }
        } else { // one pixel, taps tg * 16 .. + 15 = two chunks
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const float x[8] = {breg[8 * h], breg[8 * h + 1], breg[8 * h + 2], breg[8 * h + 3], breg[8 * h + 4], breg[8 * h + 5], breg[8 * h + 6], breg[8 * h + 7]};
                int hi[4], mid[4], lo[4];
                splitn<8, NPL>(x, hi, mid, lo);
                const int off = b_lds_off(pl_, tg * 2 + h);
                *(v4i *)(bp + off) = (v4i){hi[0], hi[1], hi[2], hi[3]};
                *(v4i *)(bp + BPLANE + off) = (v4i){mid[0], mid[1], mid[2], mid[3]};
                if (NPL == 3) *(v4i *)(bp + 2 * BPLANE + off) = (v4i){lo[0], lo[1], lo[2], lo[3]};
            }
        }
        const int aoff = a_lds_off(arow, akc >> 3) + (akc & 7) * 2; // AE bf16 = AE * 2 bytes inside the 16-byte chunk
#pragma unroll
        for (int pl = 0; pl < NPL; pl++) {
            if (AE == 8) *(v4i *)(st + pl * APLANE + aoff) = (v4i){areg[pl][0], areg[pl][1 % AD], areg[pl][2 % AD], areg[pl][3 % AD]};
            else if (AE == 4) *(int2 *)(st + pl * APLANE + aoff) = make_int2(areg[pl][0], areg[pl][1 % AD]);
            else *(int *)(st + pl * APLANE + aoff) = areg[pl][0];
        }
    };

    // accumulators start at the bias.  The MFMA takes the PIXELS as its A operand and the weights as B (D = X W^T), so a lane holds
    // pixels 4 * (lane / 16) + j of an MFMA tile for channel lane % 16: four consecutive floats of a channel row -- one 16-byte
    // store (the other way round a tile was 4 dword stores per lane: 64 store instructions per lane and 256-pixel tile)
    const int fr = lane & 15, fc = lane >> 4;
    v4f acc[MI][NI], bias4[MI];
#pragma unroll
    for (int a = 0; a < MI; a++) {
        if (RECOUT) { // the lane's four channels: 4 * fc .. + 3 of channel tile a
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int oc = oc0 + wm * TM + a * 16 + fc * 4 + j;
                bias4[a][j] = p.bias && oc < p.out_c ? p.bias[oc] : 0.f;
            }
        } else {
            const int oc = oc0 + wm * TM + a * 16 + fr;
            // (fz: the second convolution's channels sit BM / 2 further up; either convolution may have no bias -- ADVICE r5: a null bias2
            // minus BM / 2 is not null)
            const float *bsel = fz && oc >= BM / 2 ? (g.bias2 ? g.bias2 - BM / 2 : nullptr) : p.bias;
            const float b = bsel && oc < out_c_eff ? bsel[oc] : 0.f;
            bias4[a] = (v4f){b, b, b, b};
        }
#pragma unroll
        for (int c = 0; c < NI; c++) acc[a][c] = bias4[a];
    }

#ifdef SPLIT_STAMPS
    unsigned long long st_acc[4] = {0, 0, 0, 0}, st_last = 0, st_steps = 0;
#endif
    // one K step: the MFMAs of LDS stage `buf`, and the operands in (breg, areg) split into the other stage (last read in the
    // previous step, every wave is past that step's barrier; after the last step it receives stale registers nobody reads --
    // unconditional, so that it shares the MFMAs' basic block and the scheduler can interleave the two)
    auto step = [&](int buf, const v4i (&breg)[4], const int (&areg)[NPL][AD], const unsigned meta, const unsigned smeta) __attribute__((always_inline)) {
        const int8_t *ap = lds + buf * STAGE, *bp = ap + NPL * APLANE;
        bf16x8 af[NPL][MI], bf[NPL][NI];
#pragma unroll
        for (int a = 0; a < MI; a++)
#pragma unroll
            for (int pl = 0; pl < NPL; pl++)
                af[pl][a] = SPLIT_ABL & 8 ? __builtin_bit_cast(bf16x8, (v4i){buf, a, pl, lane})
                                          : __builtin_bit_cast(bf16x8, *(const v4i *)(ap + pl * APLANE + a_lds_off(wm * TM + a * 16 + fr, fc)));
#pragma unroll
        for (int pl = 0; pl < NPL; pl++)
#pragma unroll
            for (int c = 0; c < NI; c++)
                bf[pl][c] = SPLIT_ABL & 8 ? __builtin_bit_cast(bf16x8, (v4i){buf, c, pl, lane})
                                          : __builtin_bit_cast(bf16x8, *(const v4i *)(bp + pl * BPLANE + b_lds_off(wn * TN + c * 16 + fr, fc)));
        if (SPLIT_ABL & 1) {
#pragma unroll
            for (int pl = 0; pl < NPL; pl++) {
#pragma unroll
                for (int a = 0; a < MI; a++) asm volatile("" ::"v"(af[pl][a]));
#pragma unroll
                for (int c = 0; c < NI; c++) asm volatile("" ::"v"(bf[pl][c]));
            }
        }
        // the piece products of a product, smallest terms first
#pragma unroll
        for (int a = 0; a < (SPLIT_ABL & 1 ? 0 : MI); a++)
#pragma unroll
            for (int c = 0; c < NI; c++) {
                if (NPL == 3) {
                    acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0][c], af[NPL - 1][a], acc[a][c], 0, 0, 0); // lo * hi
                    acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[NPL - 1][c], af[0][a], acc[a][c], 0, 0, 0); // hi * lo
                    acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[1][c], af[1][a], acc[a][c], 0, 0, 0);       // mid * mid
                }
                if (RECOUT) { // (NPL == 2) weights are the A operand: rows = channels
                    acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][a], bf[1][c], acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][a], bf[0][c], acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][a], bf[0][c], acc[a][c], 0, 0, 0);
                    continue;
                }
                acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[1][c], af[0][a], acc[a][c], 0, 0, 0); // hi * mid
                acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0][c], af[1][a], acc[a][c], 0, 0, 0); // mid * hi
                acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0][c], af[0][a], acc[a][c], 0, 0, 0); // hi * hi
            }
        commit(buf ^ 1, breg, areg, meta, smeta);
#pragma unroll
        for (int i = 0; i < MI * NI * NPROD; i++) { // issue order: every MFMA followed by what fits in its shadow
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); // VALU
            if ((i & 7) == 7) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); // a DS write now and then
        }
        // the stage just written is visible, the stage just read is free: LDS operations only.  NOT __syncthreads(): it waits
        // vmcnt(0) as well, i.e. for the loads of step ks + 2 issued a moment ago -- a full memory latency per K step (5600
        // cycles per step whatever the MFMA count: 899 / 761 / 729 us with 6 / 4 / 3 piece products); the compiler's own
        // counted vmcnt wait sits where those registers are first used, one step later
        STAMP(1); // the step's body: fragment reads, MFMAs, split + LDS writes of the next step
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        STAMP(2); // own LDS writes landing + the wait for the other waves
    };

    const int nks = g.nks;
    fetch(0, bregs[0], aregs[0], metas[0], smetas[0]);
    commit(0, bregs[0], aregs[0], metas[0], smetas[0]);
    fetch(1, bregs[1], aregs[1], metas[1], smetas[1]);
    __syncthreads();
    // nks is even (the weight planes' rows are padded to 64 taps): two steps per iteration, no exit in between -- with a
    // `break` after the first step the compiler lost count of the loads in flight at the loop's merge points and waited
    // vmcnt(0..3) right behind every fetch, i.e. for the loads it had just issued
#ifdef SPLIT_STAMPS
    st_last = __builtin_readcyclecounter();
#endif
    // a workgroup walks a RUN of consecutive tiles: the halo rows (and, under stride 2, most of a tile's rows) it shares with the
    // next tile are then in its own XCD's L2 -- strided over the grid, neighbouring tiles ran on different XCDs at the same time
    const unsigned pt_end = (unsigned)(((unsigned long long)(blockIdx.x + 1) * g.npt) / gridDim.x);
    for (unsigned pt = pt_first; pt < pt_end; pt++) {
#ifdef SPLIT_STAMPS
        st_steps += (unsigned long long)nks;
#endif
        for (int ks = 0; ks < nks; ks += 2) {
            int kq = ks + 2;
            if (kq >= nks) { kq = 0; setup(pt + 1 < pt_end ? pt + 1 : g.npt); } // the last two steps of a tile fetch the first two of the next
            // the scheduler fence keeps a fetch's six loads AHEAD of the step's MFMAs: left alone, the compiler sinks them (with
            // their scalar address arithmetic) to the end of the step, 400 cycles before the next step needs them -- a memory
            // latency exposed per step (vmcnt(1..3) in the middle of the MFMAs)
            STAMP(3); // loop overhead, and the tile epilogue when a tile ended
            fetch(kq, bregs[0], aregs[0], metas[0], smetas[0]);
            __builtin_amdgcn_sched_barrier(0);
            STAMP(0); // the fetch: address arithmetic and six load issues
            step(0, bregs[1], aregs[1], metas[1], smetas[1]);
            STAMP(3);
            fetch(kq + 1, bregs[1], aregs[1], metas[1], smetas[1]);
            __builtin_amdgcn_sched_barrier(0);
            STAMP(0);
            step(1, bregs[0], aregs[0], metas[0], smetas[0]);
        }
        // store: a lane writes its 4 consecutive pixels of one channel row (16 bytes when the map size allows: hw % 4 == 0 keeps
        // the four in one frame and the address aligned), 4 lanes = 64 contiguous bytes, 16 channel rows per instruction
        const unsigned p0 = pt * S_BN;
        const bool vec4 = (hw & 3u) == 0u;
        if (RECOUT) {
            // a lane holds channels 4 fc .. + 3 (of channel tile a) of pixel fr (of pixel tile c): its two hi dwords and two mid dwords.
            // v_permlane16_swap between lane rows fc, fc ^ 1 (every lane takes part: before any branch): the even row ends with the
            // mid pieces of all 8 channels of the record, the odd row with the hi pieces -- one 16-byte store per lane, 32 contiguous
            // bytes per pixel, 512 per chunk and instruction
#pragma unroll
            for (int c = 0; c < NI; c++) {
                const unsigned px = p0 + (unsigned)(wn * TN + c * 16 + fr);
                const bool pok = px < g.total_pix;
                const unsigned f = sdiv(pok ? px : 0u, g.dhw), rem = (pok ? px : 0u) - f * hw;
                char *ob = (char *)p.out + (size_t)f * p.out_stride + (size_t)rem * 32u + ((fc & 1) ? 0u : 16u);
#pragma unroll
                for (int a = 0; a < MI; a++) {
                    float x[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) x[j] = p.silu ? silu_fast(acc[a][c][j]) : acc[a][c][j];
                    int hi[2], mid[2], lo_[2];
                    splitn<4, 2>(x, hi, mid, lo_);
#pragma unroll
                    for (int i = 0; i < 2; i++) { // no residual of a non-finite hi (as the weights' packer and conv_f32_patch's split)
                        const float h0 = __int_as_float(hi[i] << 16), h1 = __int_as_float(hi[i] & (int)0xffff0000);
                        mid[i] = (__builtin_isfinite(h0) ? mid[i] & 0xffff : 0) | (__builtin_isfinite(h1) ? mid[i] & (int)0xffff0000 : 0);
                    }
                    const auto s0 = __builtin_amdgcn_permlane16_swap((unsigned)mid[0], (unsigned)hi[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap((unsigned)mid[1], (unsigned)hi[1], false, false);
                    const int chunk = (oc0 + wm * TM + a * 16) / 8 + (fc >> 1);
                    if (pok && chunk * 8 < p.out_c) *(v4i *)(ob + (size_t)chunk * hw * 32u) = (v4i){(int)s0[0], (int)s1[0], (int)s0[1], (int)s1[1]};
                    acc[a][c] = bias4[a];
                }
            }
            continue;
        }
#pragma unroll
        for (int c = 0; c < NI; c++) {
            const unsigned px = p0 + (unsigned)(wn * TN + c * 16 + fc * 4);
            if (vec4) {
                if (px < g.total_pix) { // (total_pix % 4 == 0 here: all four pixels or none)
                    const unsigned f = sdiv(px, g.dhw), rem = px - f * hw;
                    float *out = (float *)((char *)p.out + (size_t)f * p.out_stride);
                    const float *addp = p.add ? (const float *)((const char *)p.add + (size_t)f * p.add_stride) : nullptr;
                    float *out2v = fz ? (float *)((char *)g.out2 + (size_t)f * p.out_stride) - (size_t)(BM / 2) * hw : out; // channel oc of the second = plane oc - BM / 2
#pragma unroll
                    for (int a = 0; a < MI; a++) {
                        const int oc = oc0 + wm * TM + a * 16 + fr;
                        if (SPLIT_ABL & 32) asm volatile("" ::"v"(acc[a][c]));
                        else if (oc < out_c_eff) {
                            v4f r = acc[a][c];
                            if (p.silu && !(SPLIT_ABL & 16)) {
#pragma unroll
                                for (int j = 0; j < 4; j++) r[j] = NPL == 2 ? silu_fast(r[j]) : silu_split(r[j]);
                            }
                            if (addp) r += *(const v4f *)(addp + (size_t)oc * hw + rem);
                            *(v4f *)((fz && wm * TM + a * 16 >= BM / 2 ? out2v : out) + (size_t)oc * hw + rem) = r;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (px + j >= g.total_pix) continue;
                    const unsigned f = sdiv(px + j, g.dhw), rem = px + j - f * hw;
                    float *out = (float *)((char *)p.out + (size_t)f * p.out_stride);
                    const float *addp = p.add ? (const float *)((const char *)p.add + (size_t)f * p.add_stride) : nullptr;
                    float *out2v = fz ? (float *)((char *)g.out2 + (size_t)f * p.out_stride) - (size_t)(BM / 2) * hw : out;
#pragma unroll
                    for (int a = 0; a < MI; a++) {
                        const int oc = oc0 + wm * TM + a * 16 + fr;
                        if (SPLIT_ABL & 32) asm volatile("" ::"v"(acc[a][c][j]));
                        else if (oc < out_c_eff) {
                            const float r = p.silu && !(SPLIT_ABL & 16) ? (NPL == 2 ? silu_fast(acc[a][c][j]) : silu_split(acc[a][c][j])) : acc[a][c][j];
                            (fz && wm * TM + a * 16 >= BM / 2 ? out2v : out)[(size_t)oc * hw + rem] = addp ? r + addp[(size_t)oc * hw + rem] : r;
                        }
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < MI; a++) acc[a][c] = bias4[a];
        }
    }
#ifdef SPLIT_STAMPS
    if (lane == 0) {
        for (int i = 0; i < 4; i++) atomicAdd(&split_stamp_sums[i], st_acc[i]);
        atomicAdd(&split_stamp_sums[4], st_steps);
    }
#endif
}

static unsigned long g_split_launches = 0; // launches of conv_f32_split since load (tests: did a shape take this kernel or the fallback?)
extern "C" unsigned long mhip_conv_f32_split_launches(void) { return g_split_launches; }

// kernel row length in the packed K space: an odd kernel width under stride 2 gets one zero column (taps come in pairs there)
static int split_kwp(int kw, int stride_w) { return stride_w == 2 && kw > 1 && (kw & 1) ? kw + 1 : kw; }

template <int BM, int WM, int WN, int GATHER, int NPL, bool RECOUT = false>
static int launch_split(const mhip_conv_f32_t *p, split_args_t g) {
    auto kern = conv_f32_split<BM, WM, WN, GATHER, NPL, RECOUT>;
    const size_t ldsb = 2 * NPL * (size_t)(BM * 64 + S_BN * 64);
    static int slots = 0; // workgroups the device holds at once (per instantiation)
    if (!slots) {
        hipDeviceProp_t prop;
        int dev = 0, occ = 0;
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess ||
            hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, S_NT, ldsb) != hipSuccess)
            return mhip_check(hipErrorUnknown, "conv_f32_split occupancy query");
        slots = (occ > 0 ? occ : 1) * prop.multiProcessorCount;
    }
    const bool fz = g.noc1 == 0xffffffffu; // the pair in one channel tile
    const unsigned noc1 = fz ? 1u : (unsigned)((p->out_c + BM - 1) / BM), noc = g.noc1 && !fz ? 2 * noc1 : noc1;
    if (noc > 65535u) return -2;
    if (g.noc1 && !fz) g.noc1 = noc1;
    // pixel tiles per workgroup: as even as the slots allow (every workgroup walks ceil(npt / gx) tiles or one fewer)
    unsigned gx = (unsigned)slots / noc;
    if (gx < 1) gx = 1;
    if (gx > g.npt) gx = g.npt;
    const unsigned per = (g.npt + gx - 1) / gx;
    gx = (g.npt + per - 1) / per;
    // (x extent a multiple of 8 when there are several channel tiles: the tiles of one pixel run then share an XCD and its L2 -- workgroups are
    // dealt to the XCDs round-robin in dispatch order, x fastest; conv_f32_patch.hip fpatch_grid_x has the measurement)
    if (noc > 1 && gx >= 8) {
        const unsigned up = (gx + 7) & ~7u, cap = (unsigned)slots / noc;
        gx = up <= cap ? up : (gx & ~7u);
    }
    g.per = per;
    hipLaunchKernelGGL(kern, dim3(gx, noc), dim3(S_NT), ldsb, mhip_stream_native(), *p, g);
    g_split_launches++;
    return mhip_check(hipGetLastError(), "conv_f32_split");
}
template <int GATHER, int NPL>
static int launch_split_bm(const mhip_conv_f32_t *p, const split_args_t &g) {
    if (g.noc1 == 0xffffffffu) { // a pair in one channel tile: BM == 2 * out_c (checked by the caller)
        if (p->out_c == 64 && NPL == 2) return launch_split<128, 2, 4, GATHER, 2>(p, g);
        if (p->out_c == 32) return launch_split<64, 1, 8, GATHER, NPL>(p, g);
        return launch_split<32, 1, 8, GATHER, NPL>(p, g);
    }
    if (p->out_rec) { // (two pieces only: checked by the caller)
        if (p->out_c > 64) return launch_split<128, 2, 4, GATHER, 2, true>(p, g);
        if (p->out_c > 32) return launch_split<64, 1, 8, GATHER, 2, true>(p, g);
        return launch_split<32, 1, 8, GATHER, 2, true>(p, g);
    }
    if (p->out_c > 64 && NPL == 2) return launch_split<128, 2, 4, GATHER, 2>(p, g); // (three planes: the 128-row tile spills)
    if (p->out_c > 32) return launch_split<64, 1, 8, GATHER, NPL>(p, g);
    return launch_split<32, 1, 8, GATHER, NPL>(p, g);
}
template <int GATHER>
static int launch_split_npl(const mhip_conv_f32_t *p, const split_args_t &g) { // use_mfma 3: two pieces, three products; 2: three, six
    return p->use_mfma == 3 ? launch_split_bm<GATHER, 2>(p, g) : launch_split_bm<GATHER, 3>(p, g);
}

// -2: not a shape this kernel takes (the caller falls back to conv_f32_mfma), else the launch result
static int try_split(const mhip_conv_f32_t *p, const mhip_conv_f32_t *q);
static unsigned long g_pair_launches = 0;
extern "C" unsigned long mhip_conv_f32_pair_launches(void) { return g_pair_launches; }
int conv_f32_try_split(const mhip_conv_f32_t *p) { return try_split(p, nullptr); }
// two convolutions over the same input, same geometry and channel count (no residual, no record output), as ONE grid; -2 = not eligible
extern "C" int mhip_conv_f32_pair(const mhip_conv_f32_t *a, const mhip_conv_f32_t *b) {
    if (!a || !b || !a->w_split || !b->w_split || a->w_patch || b->w_patch || a->use_mfma < 2 || a->use_mfma != b->use_mfma) return -2;
    if (a->in != b->in || a->in_stride != b->in_stride || a->frames != b->frames || a->in_h != b->in_h || a->in_w != b->in_w || a->in_c != b->in_c ||
        a->out_h != b->out_h || a->out_w != b->out_w || a->out_c != b->out_c || a->kh != b->kh || a->kw != b->kw || a->stride_h != b->stride_h ||
        a->stride_w != b->stride_w || a->pad_top != b->pad_top || a->pad_left != b->pad_left || a->silu != b->silu || a->out_stride != b->out_stride)
        return -2;
    if (a->add || b->add || a->in_rec || b->in_rec || a->out_rec || b->out_rec || a->out == b->out || a->k_limit != b->k_limit || a->k_limit_required != b->k_limit_required) return -2;
    const int rc = try_split(a, b);
    if (rc == 0) g_pair_launches++;
    return rc;
}
static int try_split(const mhip_conv_f32_t *p, const mhip_conv_f32_t *q) {
    if (!p->w_split || p->in_rec) return -2;
    if (p->out_rec && (p->use_mfma != 3 || p->add || (p->out_c & 7))) return -2;
    const long hw = (long)p->out_h * p->out_w, total = hw * p->frames;
    const int kwp = split_kwp(p->kw, p->stride_w);
    const long K = (long)p->in_c * p->kh * kwp;
    // k_limit: input channels from there on are exact zeros (the planner's proof): the K loop covers the others only; the weight planes keep
    // their full row length
    const long Kl = p->k_limit > 0 && p->k_limit < p->in_c ? (long)p->k_limit * p->kh * kwp : K;
    const size_t in_bytes = (size_t)(p->frames - 1) * p->in_stride + (size_t)p->in_c * p->in_h * p->in_w * 4;
    // 32-bit buffer offsets over the whole input (all frames); validity bit tables per kernel row / column
    if (total > 0x7fffffffL - S_BN || K > 0x7fffffffL - S_BK || in_bytes > 0xfffffff0ull || p->kh > 32 || kwp > 32) return -2;
    split_args_t g;
    g.total_pix = (unsigned)total; g.npt = (unsigned)((total + S_BN - 1) / S_BN); g.in_bytes = (unsigned)in_bytes;
    g.K = (int)Kl; g.kp = (int)((K + 63) / 64 * 64) + 64; g.nks = (int)((Kl + 63) / 64 * 64) / 32; g.kwp = kwp; g.oc_pad = (p->out_c + 127) / 128 * 128;
    g.dhw = make_sdiv((unsigned)hw); g.dow = make_sdiv((unsigned)p->out_w); g.dtaps = make_sdiv((unsigned)(p->kh * kwp)); g.dkwp = make_sdiv((unsigned)kwp);
    // a pair: in ONE channel tile where both fit one (2 * out_c = 32 | 64 | 128: the wave tiles then split between the two at a multiple of
    // 16 channels), else as two runs of channel tiles (the launcher fills in the count)
    const bool one_tile = q && (p->out_c == 16 || p->out_c == 32 || (p->out_c == 64 && p->use_mfma == 3));
    g.noc1 = q ? (one_tile ? 0xffffffffu : 1u) : 0u;
    g.w_split2 = q ? q->w_split : nullptr; g.bias2 = q ? q->bias : nullptr; g.out2 = q ? q->out : nullptr;
    // 16-byte gathers: 4 pixels x 1 tap (stride 1) or 2 pixels x 2 taps (stride 2) per load; pixel groups must not cross map rows
    if (p->stride_w == 1 && p->out_w % 4 == 0 && p->pad_left <= 1 && p->kw <= 8 && p->in_w >= 4) return launch_split_npl<1>(p, g);
    if (p->stride_w == 2 && p->out_w % 2 == 0 && p->pad_left <= 1 && kwp % 2 == 0 && kwp <= 16 && p->in_w >= 4) return launch_split_npl<2>(p, g);
    if (kwp != p->kw) return -2; // (the padded K space is GATHER 2's; nothing else reads it)
    return launch_split_npl<0>(p, g);
}

// Geometry test for the planner (rec_pairs): which gather form conv_f32_split takes for this shape -- 1 / 2 (16-byte gathers, stride 1 / 2),
// 0 (one dword per tap) -- or -1 when try_split declines it (the same conditions, nothing batch-dependent)
extern "C" int mhip_conv_f32_split_takes(int out_c, int in_c, int kh, int kw, int stride_h, int stride_w, int pad_left, int in_w, int out_w) {
    if (out_c <= 0 || in_c <= 0 || kh <= 0 || kw <= 0 || stride_h <= 0 || stride_w <= 0) return -1;
    const int kwp = split_kwp(kw, stride_w);
    if ((long)in_c * kh * kwp > 0x7fffffffL - S_BK || kh > 32 || kwp > 32) return -1;
    if (stride_w == 1 && out_w % 4 == 0 && pad_left <= 1 && kw <= 8 && in_w >= 4) return 1;
    if (stride_w == 2 && out_w % 2 == 0 && pad_left <= 1 && kwp % 2 == 0 && kwp <= 16 && in_w >= 4) return 2;
    return kwp != kw ? -1 : 0;
}

static uint16_t bf16_rn(float x) { // round to nearest even (NaN kept a NaN)
    uint32_t b;
    memcpy(&b, &x, 4);
    if ((b & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((b >> 16) | 0x40u);
    return (uint16_t)((b + 0x7fffu + ((b >> 16) & 1u)) >> 16);
}
static float bf16_val(uint16_t h) {
    const uint32_t b = (uint32_t)h << 16;
    float f;
    memcpy(&f, &b, 4);
    return f;
}
extern "C" size_t mhip_conv_f32_split_pack(int out_c, int in_c, int kh, int kw, int stride_w, int planes, const float *w, void *out) {
    if (out_c <= 0 || in_c <= 0 || kh <= 0 || kw <= 0 || planes < 2 || planes > 3) return 0;
    const int kwp = split_kwp(kw, stride_w);
    const size_t K = (size_t)in_c * kh * kwp, kp = (K + 63) / 64 * 64 + 64, ocp = ((size_t)out_c + 127) / 128 * 128;
    const size_t bytes = (size_t)planes * ocp * kp * 2;
    if (!w || !out) return bytes;
    uint16_t *o = (uint16_t *)out;
    memset(o, 0, bytes);
    for (int oc = 0; oc < out_c; oc++)
        for (int ic = 0; ic < in_c; ic++)
            for (int ky = 0; ky < kh; ky++)
                for (int kx = 0; kx < kw; kx++) {
                    const float x = w[((size_t)(oc * (size_t)in_c + ic) * kh + ky) * kw + kx];
                    const uint16_t h = bf16_rn(x);
                    const float hv = bf16_val(h);
                    const float r1 = hv - hv == 0.0f ? x - hv : 0.0f; // exact; hi not finite (|x| >= 2^128 - 2^119, inf, NaN): no residual
                    const uint16_t m = bf16_rn(r1);
                    const float r2 = r1 - bf16_val(m); // exact, a bf16 value
                    const size_t k = ((size_t)ic * kh + ky) * kwp + kx;
                    o[(size_t)oc * kp + k] = h;
                    o[ocp * kp + (size_t)oc * kp + k] = m;
                    if (planes == 3) o[2 * ocp * kp + (size_t)oc * kp + k] = bf16_rn(r2);
                }
    return bytes;
}
