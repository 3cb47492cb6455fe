// conv_f32_patch.hip -- float32 k x k convolution (NCHW / OIHW, reference src/mars/mxu_conv.c:673-710) on the bf16 matrix
// cores with the INPUT STAGED ONCE PER TILE: round 5, verdict item 2.  mhip_conv_f32_t.use_mfma == 3 ("bf16x3": every operand
// cut exactly into two bf16 pieces, three piece products per product; conv_f32_split.hip explains the arithmetic) for the
// layers whose kernel has >= 8 taps (3x3, 5x5, 6x6 ...), stride 1 or 2, in_c a multiple of 8.
//
// What was wrong with the implicit-GEMM form (conv_f32_split) on these layers, by its own stamps (profiles/r04_experiments.md
// section 3): a K step of ~4100 cycles held 1536 cycles of matrix work; the rest was the im2col gather -- every input element
// fetched through L1 once per TAP (9 x for a 3x3) -- and the bf16 split repeated on every one of those fetches (~150 vector
// instructions per step against the 96 slots the MFMAs leave).  Here
//   * a workgroup's tile is 256 consecutive output pixels of a STRIP ORDER (strips of SW output columns, rows inside a strip,
//     strips inside a frame, frames stacked): wide maps get 2-D tiles (8 x 32, 16 x 16), narrow maps (40 / 20 wide) whole-row
//     tiles that run on into the next frame, so every MFMA row holds a real pixel on every map size;
//   * the tile's input PATCH (its rows and columns plus the halo) is fetched 8 channels (one CHUNK) at a time by 16-byte
//     loads, ONCE, split ONCE into its two bf16 pieces (v_cvt_pk_bf16_f32 + exact residual) and written channels-last into LDS:
//     a pixel's record is 32 bytes = [8 x hi | 8 x mid] (the halves swapped on every other group of 8 pixels: bank
//     spreading).  The patch ring holds two chunks;
//   * the K stream is a sequence of UNITS (tap, chunk) = 8 K-elements = one 16-byte LDS read per pixel row; an MFMA K step
//     (32 elements) is four consecutive units, each lane group (lane / 16) reading its own unit at patch pixel
//     Pbase(pixel) + toff(unit): no im2col anywhere, the nine taps are nine LDS offsets.  Units run on across chunk
//     boundaries (9 taps do not divide by 4), so a step may read both ring slots;
//   * weights: split into hi / mid on the host in exactly that K order (mhip_conv_f32_patch_pack), staged per K step through a
//     two-stage LDS pair by the register path of conv_f32_split (16 bytes per thread and plane);
//   * WHEN a chunk is fetched (8 loads per thread) and committed (split + 8 LDS writes) is a table the host derives from the
//     unit stream (fpatch_schedule): chunk g is written one step after the last step that reads chunk g - 2, its loads were
//     issued when chunk g - 1 was written (>= 2 steps earlier); the kernel walks pixel tiles persistently and the stream
//     runs through tile boundaries (the next tile's first chunks are fetched during the last steps of this one).
// Everything index-shaped (geometry, unit table, schedule, weight order) is host code shared with the packer and checked on
// the CPU against a direct convolution by an emulation of this kernel's addressing (tests/test_host_pack.py).
// Results: inside north_star's 1e-4 (same three piece products as conv_f32_split; the K order differs, so not bit-equal to it).
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

#define P_BN 256   // pixels per tile
#define P_NT 512   // threads per workgroup
#define P_NB 2     // patch ring slots (chunks)
#define P_PRCAP 96 // patch rows, at most
#define P_MAGIC 0x35504650

struct pdiv_t {
    unsigned m, s1, s2;
};
__host__ __device__ __forceinline__ unsigned pdiv(unsigned n, const pdiv_t d) {
#ifdef __HIP_DEVICE_COMPILE__
    const unsigned q = __umulhi(d.m, n);
#else
    const unsigned q = (unsigned)(((unsigned long long)d.m * n) >> 32);
#endif
    return (q + ((n - q) >> d.s1)) >> d.s2;
}
static pdiv_t make_pdiv(unsigned d) {
    pdiv_t r;
    unsigned l = 0;
    while ((1ull << l) < d) l++;
    r.m = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    r.s1 = l < 1 ? l : 1;
    r.s2 = l > 0 ? l - 1 : 0;
    return r;
}

// geometry of one layer (host-derived; the ints before the dividers are also what mhip_conv_f32_patch_geom reports)
struct fpatch_geom_t {
    int s, kh, kw, pad;        // stride (both axes), kernel, padding (top == left)
    int C, nchunk, U;          // input channels, 8-channel chunks, real units (taps) per chunk
    int SW, nstrips;           // strip width (output pixels), strips per frame
    int H_in, W_in, H_out, W_out;
    int HV;                    // virtual input rows of a strip segment = (H_out - 1) * s + kh
    int PR, PWP, PWH, dx;      // patch rows, row pitch (pixels, multiple of 8), half pitch (stride 2), column of tap 0 of strip column 0
    int slotpix;               // pixels per ring slot
    int nsteps;                // K steps per tile (even; dummy units pad the stream)
    int ngrp, nitems;          // 4-column groups per patch row, fetch items per chunk (threads that fetch)
    int BM, kp, oc_pad;        // channel tile, weight row length (bf16 elements), weight rows per plane
    int tab_ints;              // ints of the table block in front of the weight planes
    int ndummy;                // dummy units at the end of a tile's stream (table entry -1: multiply zeros)
    int woff, poff, lds_bytes; // LDS byte offsets of the weight stages and the patch ring, total
    unsigned total_pix, ntiles, nsegs, in_bytes, per;
    pdiv_t dSW, dHo, dHV, dNS, dgrp;
};

__device__ __forceinline__ int pa_lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 2)) << 4); }
__device__ __forceinline__ float psilu_fast(float v) { // as conv_f32_split's three-product mode: v_exp_f32 / v_rcp_f32
    const float e = __builtin_amdgcn_exp2f(v * -1.44269504088896341f);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}

// eight floats (one pixel, 8 channels) -> 4 dwords of hi, 4 of mid.  hi = bf16(x) (round to nearest even), mid = bf16(x - hi); the
// subtraction is exact.  Not finite (hi = +-inf or NaN, i.e. |x| >= 2^128 - 2^119 or x not finite): mid = 0, so that the product sums
// see the inf / NaN once instead of the NaN an inf - inf residual would be
__device__ __forceinline__ void psplit8(const float (&x)[8], v4i &hi, v4i &mid) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const f32x2 v = {x[2 * i], x[2 * i + 1]};
        const int h = __builtin_bit_cast(int, __builtin_convertvector(v, bf16x2));
        const float h0 = __int_as_float(h << 16), h1 = __int_as_float(h & (int)0xffff0000);
        f32x2 r = {v[0] - h0, v[1] - h1};
        r[0] = __builtin_isfinite(h0) ? r[0] : 0.0f;
        r[1] = __builtin_isfinite(h1) ? r[1] : 0.0f;
        hi[i] = h;
        mid[i] = __builtin_bit_cast(int, __builtin_convertvector(r, bf16x2));
    }
}

// BM = output channels per workgroup; waves WM (channels) x WN (pixels), WM * WN == 8
template <int BM, int WM, int WN>
__global__ __launch_bounds__(P_NT, 2) void conv_f32_patch(const mhip_conv_f32_t p, const fpatch_geom_t g, const int *__restrict__ tabs,
                                                          const int8_t *__restrict__ wpl) {
    constexpr int TM = BM / WM, TN = P_BN / WN;
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int APLANE = BM * 64;        // one weight plane of a K step
    constexpr int WSTAGE = 2 * APLANE;     // hi + mid
    constexpr int AE = BM * 32 / P_NT;     // weight elements per thread, plane and step: 8 | 4 | 2
    constexpr int ATPR = 32 / AE, AD = AE / 2;
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    int *dutab = (int *)lds;                                 // [nsteps][4] patch-pixel offset of the step's four units (slot included)
    int *sched = dutab + g.nsteps * 4;                       // [nsteps] 0 | 1 + chunk to commit at the top of the step
    int2 *rowtab = (int2 *)(lds + g.woff - 2 * P_PRCAP * 8); // [2][P_PRCAP] (row byte offset | ~0, first column x_al)
    int8_t *wst = lds + g.woff;
    int8_t *patch = lds + g.poff;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv % WM, wn = wv / WM;
    const int fr = lane & 15, fc = lane >> 4;
    const int oc0 = (int)blockIdx.y * BM;
    const unsigned hw = (unsigned)(g.H_out * g.W_out);
    const unsigned plane_bytes = (unsigned)(g.H_in * g.W_in) * 4u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, (int)g.in_bytes, 0x00020000);

    for (int i = tid; i < g.nsteps * 5; i += P_NT) dutab[i] = tabs[i]; // dutab and sched are contiguous in both places

    // ---- this thread's fetch item: patch row `ir`, columns 4 * igq .. + 3, all 8 channels of a chunk
    const bool has_item = tid < g.nitems;
    const int ir = (int)pdiv((unsigned)tid, g.dgrp), igq = tid - ir * g.ngrp;
    int pitem[4]; // patch pixel (inside a slot) of the item's four columns
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int v = 4 * igq + i;
        pitem[i] = ir * g.PWP + (g.s == 2 ? (v >> 1) + (v & 1) * g.PWH : v);
    }
    // first virtual row of tile t
    auto tile_v0 = [&](unsigned t) __attribute__((always_inline)) {
        const unsigned R0 = pdiv(t * P_BN, g.dSW), seg0 = pdiv(R0, g.dHo);
        return (int)(seg0 * (unsigned)g.HV + (R0 - seg0 * (unsigned)g.H_out) * (unsigned)g.s);
    };
    auto fill_rowtab = [&](unsigned t) __attribute__((always_inline)) { // source of every patch row of tile t (threads < PR)
        if (tid < g.PR) {
            const unsigned V = (unsigned)tile_v0(t) + (unsigned)tid;
            const unsigned seg = pdiv(V, g.dHV), f = pdiv(seg, g.dNS), st = seg - f * (unsigned)g.nstrips;
            const int iy = (int)(V - seg * (unsigned)g.HV) - g.pad;
            const bool ok = t < g.ntiles && seg < g.nsegs && iy >= 0 && iy < g.H_in;
            rowtab[(t & 1) * P_PRCAP + tid] = make_int2(ok ? (int)(f * (unsigned)p.in_stride + (unsigned)(iy * g.W_in) * 4u) : -1,
                                                        (int)st * g.SW * g.s - g.pad - g.dx);
        }
    };
    v4i breg[8]; // the chunk in flight: channel j, 4 columns
    auto fetch_patch = [&](unsigned t, int chunk) __attribute__((always_inline)) { // chunk of tile t (rowtab[t & 1] is in place)
        unsigned vo = 0xffffffffu;
        if (has_item) {
            const int2 rt = rowtab[(t & 1) * P_PRCAP + ir];
            const int x = rt.y + 4 * igq;
            if (rt.x != -1 && x >= 0 && x < g.W_in) vo = (unsigned)rt.x + (unsigned)x * 4u;
        }
        unsigned so = (unsigned)chunk * 8u * plane_bytes;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            breg[j] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, so, 0));
            so += plane_bytes;
        }
    };
    auto commit_patch = [&](int slot) __attribute__((always_inline)) {
        if (has_item) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float x[8] = {__int_as_float(breg[0][i]), __int_as_float(breg[1][i]), __int_as_float(breg[2][i]), __int_as_float(breg[3][i]),
                                    __int_as_float(breg[4][i]), __int_as_float(breg[5][i]), __int_as_float(breg[6][i]), __int_as_float(breg[7][i])};
                v4i hi, mid;
                psplit8(x, hi, mid);
                const int P = slot * g.slotpix + pitem[i];
                const int a = P * 32 + (((P >> 3) & 1) << 4);
                *(v4i *)(patch + a) = hi;
                *(v4i *)(patch + (a ^ 16)) = mid;
            }
        }
    };

    // ---- weights: row oc0 + tid / ATPR of both planes, elements (tid % ATPR) * AE .. of the step (as conv_f32_split)
    const int arow = tid / ATPR, akc = (tid % ATPR) * AE;
    const int8_t *wrow = wpl + ((size_t)(oc0 + arow) * g.kp + akc) * 2;
    const size_t wplane = (size_t)g.oc_pad * g.kp * 2;
    int aregs[2][2][AD];
    auto fetch_w = [&](int ks, int (&areg)[2][AD]) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            const int8_t *src = wrow + pl * wplane + (size_t)ks * 64;
            if (AE == 8) { const v4i t = *(const v4i *)src; areg[pl][0] = t[0]; areg[pl][1 % AD] = t[1]; areg[pl][2 % AD] = t[2]; areg[pl][3 % AD] = t[3]; }
            else if (AE == 4) { const int2 t = *(const int2 *)src; areg[pl][0] = t.x; areg[pl][1 % AD] = t.y; }
            else areg[pl][0] = *(const int *)src;
        }
    };
    auto commit_w = [&](int buf, const int (&areg)[2][AD]) __attribute__((always_inline)) {
        int8_t *st = wst + buf * WSTAGE;
        const int aoff = pa_lds_off(arow, akc >> 3) + (akc & 7) * 2;
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            if (AE == 8) *(v4i *)(st + pl * APLANE + aoff) = (v4i){areg[pl][0], areg[pl][1 % AD], areg[pl][2 % AD], areg[pl][3 % AD]};
            else if (AE == 4) *(int2 *)(st + pl * APLANE + aoff) = make_int2(areg[pl][0], areg[pl][1 % AD]);
            else *(int *)(st + pl * APLANE + aoff) = areg[pl][0];
        }
    };

    // ---- the compute side's view of a tile: patch pixel of tap (0, 0) of this lane's A rows, output offsets of its D rows
    int pbase[NI];
    unsigned ooff[NI]; // byte offset (frame + position inside a channel plane) of the lane's 4 result pixels, ~0 = none
    auto tile_setup = [&](unsigned t) __attribute__((always_inline)) {
        const int V0 = tile_v0(t);
#pragma unroll
        for (int n = 0; n < NI; n++) {
            const unsigned q = t * P_BN + (unsigned)(wn * TN + n * 16 + fr); // A operand: pixel fr of MFMA tile n
            const unsigned R = pdiv(q, g.dSW), xs = q - R * (unsigned)g.SW;
            const unsigned seg = pdiv(R, g.dHo), y = R - seg * (unsigned)g.H_out;
            const int prow = (int)(seg * (unsigned)g.HV + y * (unsigned)g.s) - V0;
            pbase[n] = q < g.total_pix ? prow * g.PWP + (int)xs : 0;
            const unsigned q4 = t * P_BN + (unsigned)(wn * TN + n * 16 + fc * 4); // D: pixels 4 fc .. + 3 of tile n, channel fr
            const unsigned R4 = pdiv(q4, g.dSW), xs4 = q4 - R4 * (unsigned)g.SW;
            const unsigned seg4 = pdiv(R4, g.dHo), y4 = R4 - seg4 * (unsigned)g.H_out;
            const unsigned f4 = pdiv(seg4, g.dNS), st4 = seg4 - f4 * (unsigned)g.nstrips;
            ooff[n] = q4 < g.total_pix ? f4 * (unsigned)p.out_stride + (y4 * (unsigned)g.W_out + st4 * (unsigned)g.SW + xs4) * 4u : 0xffffffffu;
        }
    };

    v4f acc[MI][NI], bias4[MI];
#pragma unroll
    for (int a = 0; a < MI; a++) {
        const int oc = oc0 + wm * TM + a * 16 + fr;
        const float b = p.bias && oc < p.out_c ? p.bias[oc] : 0.f;
        bias4[a] = (v4f){b, b, b, b};
#pragma unroll
        for (int c = 0; c < NI; c++) acc[a][c] = bias4[a];
    }

    const int nsteps_m2 = g.nsteps - 2;
    // one K step: the MFMAs of weight stage `buf` on the units of step t, and the weights in `areg` written into the other stage
    auto step = [&](int t, int buf, const int (&areg)[2][AD]) __attribute__((always_inline)) {
        const int8_t *ap = wst + buf * WSTAGE;
        const int e = dutab[t * 4 + fc];
        bf16x8 xh[NI], xm[NI], wh[MI], wmid[MI];
#pragma unroll
        for (int c = 0; c < NI; c++) {
            const int P = pbase[c] + (e < 0 ? 0 : e);
            const int a = P * 32 + (((P >> 3) & 1) << 4);
            xh[c] = __builtin_bit_cast(bf16x8, *(const v4i *)(patch + a));
            xm[c] = __builtin_bit_cast(bf16x8, *(const v4i *)(patch + (a ^ 16)));
        }
        if (g.ndummy && t >= nsteps_m2) { // dummy units (the stream padded to an even number of steps) multiply ZEROS: their weights are
                                          // zero, but 0 x inf would be a NaN the reference does not have (the float twins do hold infs)
            const bool dm = e < 0;
#pragma unroll
            for (int c = 0; c < NI; c++) {
                v4i h = __builtin_bit_cast(v4i, xh[c]), m = __builtin_bit_cast(v4i, xm[c]);
#pragma unroll
                for (int j = 0; j < 4; j++) { h[j] = dm ? 0 : h[j]; m[j] = dm ? 0 : m[j]; }
                xh[c] = __builtin_bit_cast(bf16x8, h); xm[c] = __builtin_bit_cast(bf16x8, m);
            }
        }
#pragma unroll
        for (int a = 0; a < MI; a++) {
            const int o = pa_lds_off(wm * TM + a * 16 + fr, fc);
            wh[a] = __builtin_bit_cast(bf16x8, *(const v4i *)(ap + o));
            wmid[a] = __builtin_bit_cast(bf16x8, *(const v4i *)(ap + APLANE + o));
        }
#pragma unroll
        for (int a = 0; a < MI; a++)
#pragma unroll
            for (int c = 0; c < NI; c++) { // pixels are the A operand: a lane ends with 4 consecutive pixels of one channel
                acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm[c], wh[a], acc[a][c], 0, 0, 0);   // mid * hi
                acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[c], wmid[a], acc[a][c], 0, 0, 0); // hi * mid
                acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[c], wh[a], acc[a][c], 0, 0, 0);   // hi * hi
            }
        commit_w(buf ^ 1, areg);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // LDS only: the loads in flight stay in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // top of step t: commit the chunk the schedule names (its loads were issued >= 2 steps ago), then fetch the next one
    auto patch_action = [&](unsigned t_tile, int t) __attribute__((always_inline)) {
        const int sc = __builtin_amdgcn_readfirstlane(sched[t]);
        if (sc) {
            const int cc = sc - 1; // chunk of this tile, or nchunk + chunk of the next
            commit_patch(cc & (P_NB - 1));
            const int nx = cc + 1;
            if (nx >= g.nchunk) fetch_patch(t_tile + 1, nx - g.nchunk);
            else fetch_patch(t_tile, nx);
        }
    };

    // ---- prologue: tables, the first two chunks of the first tile, the first weight steps
    const unsigned t_first = blockIdx.x * g.per;
    const unsigned t_end = (blockIdx.x + 1) * g.per < g.ntiles ? (blockIdx.x + 1) * g.per : g.ntiles;
    fill_rowtab(t_first);
    __syncthreads();
    // the state every later tile starts in: what the PREVIOUS tile's steps would have done for this tile's first chunks (the
    // schedule entries that name chunk nchunk + c), in order -- chunk 0's loads, then per entry: commit, fetch the next
    fetch_patch(t_first, 0);
    for (int i = 0; i < g.nsteps; i++) {
        const int sc = __builtin_amdgcn_readfirstlane(sched[i]);
        if (sc > g.nchunk) {
            const int cc = sc - 1 - g.nchunk;
            commit_patch(cc & (P_NB - 1));
            fetch_patch(t_first, cc + 1);
        }
    }
    fetch_w(0, aregs[0]);
    commit_w(0, aregs[0]);
    fetch_w(1, aregs[1]);
    __syncthreads();

    const int nsteps = g.nsteps;
    for (unsigned t = t_first; t < t_end; t++) {
        tile_setup(t);
        for (int ks = 0; ks < nsteps; ks += 2) {
            int kq = ks + 2;
            if (kq >= nsteps) kq = 0; // the last two steps of a tile fetch the weights of the next tile's first two
            if (ks == 0) fill_rowtab(t + 1); // the other table held tile t - 1's rows: its last chunk was fetched during tile t - 1;
                                             // this one is first read by a fetch at step >= 1 (nchunk >= 4), behind step 0's barrier
            patch_action(t, ks);
            fetch_w(kq, aregs[0]);
            __builtin_amdgcn_sched_barrier(0);
            step(ks, 0, aregs[1]);
            patch_action(t, ks + 1);
            fetch_w(kq + 1, aregs[1]);
            __builtin_amdgcn_sched_barrier(0);
            step(ks + 1, 1, aregs[0]);
        }
        // ---- store: a lane holds 4 consecutive pixels (one 16-byte store) of channel row fr of every MFMA tile
#pragma unroll
        for (int c = 0; c < NI; c++) {
            if (ooff[c] != 0xffffffffu) {
#pragma unroll
                for (int a = 0; a < MI; a++) {
                    const int oc = oc0 + wm * TM + a * 16 + fr;
                    if (oc < p.out_c) {
                        v4f r = acc[a][c];
                        if (p.silu) {
#pragma unroll
                            for (int j = 0; j < 4; j++) r[j] = psilu_fast(r[j]);
                        }
                        const size_t o = (size_t)ooff[c] + (size_t)oc * hw * 4u;
                        if (p.add) r += *(const v4f *)((const char *)p.add + o); // (add_stride == out_stride: checked by the launcher)
                        *(v4f *)((char *)p.out + o) = r;
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < MI; a++) acc[a][c] = bias4[a];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side: geometry, unit table, schedule, weight image

// the K stream of one tile: units in (chunk, tap) order, padded with dummy units (-1) to an even number of 4-unit steps
static int fpatch_units(int nchunk, int U, int *units /* [nsteps * 4] chunk of each unit or -1 */, int cap) {
    const int real = nchunk * U;
    int nsteps = (real + 3) / 4;
    if (nsteps & 1) nsteps++;
    if (nsteps * 4 > cap) return -1;
    for (int i = 0; i < nsteps * 4; i++) units[i] = i < real ? i / U : -1;
    return nsteps;
}
// commit step (inside a tile's step numbering, chunk indices continuing into the next tile) of every chunk: chunk g is written
// at the top of the step after the last one that reads chunk g - 2 (same slot), at least 2 steps after chunk g - 1 (whose commit
// issued g's loads), and at least one step before g's first reader.  Returns 0 if no such schedule exists.
static int fpatch_schedule(int nchunk, int U, int nsteps, int *sched /* [nsteps] */) {
    const int NB = P_NB;
    if (nchunk < NB || nchunk % NB) return 0;
    // first / last reading step of chunk c of the tile (global: + nsteps per tile)
    auto first_read = [&](long gch) { const long t = gch / nchunk, c = gch % nchunk; return t * nsteps + (c * U) / 4; };
    auto last_read = [&](long gch) { const long t = gch / nchunk, c = gch % nchunk; return t * nsteps + ((c + 1) * U - 1) / 4; };
    for (int i = 0; i < nsteps; i++) sched[i] = 0;
    // chunks 0, 1 of the first tile are committed by the prologue ("step -1"); steady state from chunk NB on.  The pattern must
    // repeat from tile to tile: simulate three tiles and keep the commits that fall into tile 1's steps
    long prev = -1;
    for (long gch = NB; gch < 3L * nchunk + NB; gch++) {
        long c = last_read(gch - NB) + 1;
        if (c < prev + 2) c = prev + 2;
        if (c > first_read(gch) - 1) return 0;
        prev = c;
        if (c >= nsteps && c < 2L * nsteps) {
            const long rel = gch - nchunk; // chunk index relative to tile 1: may reach into tile 2 (>= nchunk)
            if (rel < 0 || rel >= 2L * nchunk || sched[c - nsteps]) return 0;
            sched[c - nsteps] = (int)rel + 1;
        }
    }
    // the same commits, seen from tile 0, must be what tile 1 shows (periodicity): chunk g of tile 0 at step c <=> chunk g of tile 1 at c
    // (holds because first/last_read are tile-periodic and the prologue's state equals the steady state's: checked by the emulation test)
    return 1;
}

static int fpatch_geom(const mhip_conv_f32_t *p, fpatch_geom_t *g, int frames) {
    memset(g, 0, sizeof(*g));
    const int s = p->stride_w;
    if (p->stride_h != s || (s != 1 && s != 2) || p->pad_top != p->pad_left || p->pad_top < 0 || p->pad_top > 3) return 0;
    if (p->in_c < 32 || (p->in_c & 7) || p->kh * p->kw < 8 || p->kh > 7 || p->kw > 7) return 0;
    if ((p->in_w & 3) || (p->out_w & 3) || p->out_h < 1) return 0;
    // every tap of every output pixel must lie inside the virtual (padded) rows / the patch columns: SAME-style geometry
    if ((p->out_h - 1) * s + p->kh - p->pad_top > p->in_h + 3 || (p->out_w - 1) * s + p->kw - p->pad_left > p->in_w + 3) return 0;
    g->s = s; g->kh = p->kh; g->kw = p->kw; g->pad = p->pad_top;
    g->C = p->in_c; g->nchunk = p->in_c / 8; g->U = p->kh * p->kw;
    if (g->nchunk % P_NB || g->nchunk < 4) return 0; // (two row tables: a tile's last chunk must be fetched during that tile)
    g->H_in = p->in_h; g->W_in = p->in_w; g->H_out = p->out_h; g->W_out = p->out_w;
    g->HV = (p->out_h - 1) * s + p->kh;
    g->BM = p->out_c > 64 ? 128 : (p->out_c > 32 ? 64 : 32);
    g->oc_pad = (p->out_c + 127) / 128 * 128;
    int units[4096];
    g->nsteps = fpatch_units(g->nchunk, g->U, units, 4096);
    if (g->nsteps < 2) return 0;
    g->ndummy = g->nsteps * 4 - g->nchunk * g->U;
    g->kp = g->nsteps * 32 + 64; // (fetches run two steps ahead and wrap: the slack is never multiplied)
    g->tab_ints = g->nsteps * 5;
    // strip width: a divisor of out_w, multiple of 4; the one with the smallest patch (ties: the wider)
    const int woff_base = (g->nsteps * 5 * 4 + 2 * P_PRCAP * 8 + 255) & ~255;
    const int wbytes = 2 * 2 * g->BM * 64;
    int best = 0;
    for (int SW = 4; SW <= p->out_w; SW += 4) {
        if (p->out_w % SW) continue;
        const int x0 = -p->pad_left;                     // input column of tap 0 of strip column 0 (strip 0)
        const int xal = (x0 >= 0 ? x0 : x0 - 3) / 4 * 4; // floor to a multiple of 4 (the same residue for every strip: SW * s % 4 == 0)
        const int dx = x0 - xal;
        const int PWP = (dx + (SW - 1) * s + p->kw + 7) & ~7;
        const int NR = P_BN % SW == 0 ? P_BN / SW : (P_BN + SW - 2) / SW + 1; // output rows a tile can touch
        const int ncross = (NR + p->out_h - 2) / p->out_h;                    // strip / frame boundaries inside them, at most
        const int PR = (NR - 1) * s + p->kh + ncross * (p->kh - s > 0 ? p->kh - s : 0);
        if (PR > P_PRCAP) continue;
        const int nitems = PR * (PWP / 4);
        const int slotpix = (PR * PWP + 7) & ~7;
        const int lds = woff_base + wbytes + P_NB * slotpix * 32;
        if (nitems > P_NT || lds > 160 * 1024) continue;
        if (!best || PR * PWP < g->PR * g->PWP || (PR * PWP == g->PR * g->PWP && SW > g->SW)) {
            best = 1;
            g->SW = SW; g->dx = dx; g->PWP = PWP; g->PWH = PWP / 2; g->PR = PR; g->nitems = nitems; g->ngrp = PWP / 4; g->slotpix = slotpix;
            g->woff = woff_base; g->poff = woff_base + wbytes; g->lds_bytes = lds;
        }
    }
    if (!best) return 0;
    g->nstrips = p->out_w / g->SW;
    int sched[1024];
    if (g->nsteps > 1024 || !fpatch_schedule(g->nchunk, g->U, g->nsteps, sched)) return 0;
    const long total = (long)frames * p->out_h * p->out_w;
    const size_t in_bytes = (size_t)(frames - 1) * p->in_stride + (size_t)p->in_c * p->in_h * p->in_w * 4;
    if (total > 0x7fffffffL - P_BN || in_bytes > 0xfffffff0ull || (size_t)frames * p->out_stride > 0xfffffff0ull) return 0;
    g->total_pix = (unsigned)total;
    g->ntiles = (unsigned)((total + P_BN - 1) / P_BN);
    g->nsegs = (unsigned)(frames * g->nstrips);
    g->in_bytes = (unsigned)in_bytes;
    g->dSW = make_pdiv((unsigned)g->SW); g->dHo = make_pdiv((unsigned)g->H_out); g->dHV = make_pdiv((unsigned)g->HV);
    g->dNS = make_pdiv((unsigned)g->nstrips); g->dgrp = make_pdiv((unsigned)g->ngrp);
    return 1;
}

// patch-pixel offset of unit (chunk c, tap) of the K stream: ring slot + tap position (stride 2: de-interleaved columns)
static int fpatch_toff(const fpatch_geom_t *g, int c, int tap) {
    const int ky = tap / g->kw, kx = tap - ky * g->kw, v = g->dx + kx;
    return (c % P_NB) * g->slotpix + ky * g->PWP + (g->s == 2 ? (v >> 1) + (v & 1) * g->PWH : v);
}

static uint16_t pbf16_rn(float x) {
    uint32_t b;
    memcpy(&b, &x, 4);
    if ((b & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((b >> 16) | 0x40u);
    return (uint16_t)((b + 0x7fffu + ((b >> 16) & 1u)) >> 16);
}
static float pbf16_val(uint16_t h) {
    const uint32_t b = (uint32_t)h << 16;
    float f;
    memcpy(&f, &b, 4);
    return f;
}

static void shape_of(mhip_conv_f32_t *p, int out_c, int in_c, int kh, int kw, int stride, int pad, int in_h, int in_w, int out_h, int out_w) {
    memset(p, 0, sizeof(*p));
    p->out_c = out_c; p->in_c = in_c; p->kh = kh; p->kw = kw; p->stride_h = p->stride_w = stride; p->pad_top = p->pad_left = pad;
    p->in_h = in_h; p->in_w = in_w; p->out_h = out_h; p->out_w = out_w;
    p->in_stride = (size_t)in_c * in_h * in_w * 4; p->out_stride = (size_t)out_c * out_h * out_w * 4; p->frames = 1;
}

// Bytes of, and (w, out != NULL) the content of, the image conv_f32_patch reads: [nsteps][4] unit offsets, [nsteps] schedule, then
// two planes (hi, mid) of bf16 [oc_pad][kp] in the kernel's K order: element 8 u + j of a row = channel 8 c + j, tap of unit u =
// (chunk c, tap) (dummy units and the slack: zeros).  0 = not a shape this kernel takes.
extern "C" size_t mhip_conv_f32_patch_pack(int out_c, int in_c, int kh, int kw, int stride, int pad, int in_h, int in_w, int out_h, int out_w,
                                           const float *w, void *out) {
    mhip_conv_f32_t p;
    shape_of(&p, out_c, in_c, kh, kw, stride, pad, in_h, in_w, out_h, out_w);
    fpatch_geom_t g;
    if (out_c <= 0 || !fpatch_geom(&p, &g, 1)) return 0;
    const size_t tabb = ((size_t)g.tab_ints * 4 + 255) & ~(size_t)255, planeb = (size_t)g.oc_pad * g.kp * 2;
    const size_t bytes = tabb + 2 * planeb;
    if (!w || !out) return bytes;
    memset(out, 0, bytes);
    int *tabs = (int *)out;
    int units[4096];
    fpatch_units(g.nchunk, g.U, units, 4096);
    for (int i = 0; i < g.nsteps * 4; i++) tabs[i] = units[i] < 0 ? -1 : fpatch_toff(&g, units[i], i - units[i] * g.U);
    fpatch_schedule(g.nchunk, g.U, g.nsteps, tabs + g.nsteps * 4);
    uint16_t *hi = (uint16_t *)((char *)out + tabb), *mid = hi + (size_t)g.oc_pad * g.kp;
    for (int oc = 0; oc < out_c; oc++)
        for (int u = 0; u < g.nchunk * g.U; u++) {
            const int c = u / g.U, tap = u - c * g.U, ky = tap / kw, kx = tap - ky * kw;
            for (int j = 0; j < 8; j++) {
                const float x = w[((size_t)(oc * (size_t)in_c + c * 8 + j) * kh + ky) * kw + kx];
                const uint16_t h = pbf16_rn(x);
                const float hv = pbf16_val(h);
                const size_t k = (size_t)oc * g.kp + (size_t)u * 8 + j;
                hi[k] = h;
                mid[k] = (hv - hv == 0.0f) ? pbf16_rn(x - hv) : 0; // (x - hi exact; hi not finite: no residual)
            }
        }
    return bytes;
}

// the geometry as ints (tests / tools): s kh kw pad C nchunk U SW nstrips H_in W_in H_out W_out HV PR PWP PWH dx slotpix nsteps ngrp
// nitems BM kp oc_pad tab_ints ndummy woff poff lds_bytes; returns how many were written (0 = not a shape this kernel takes)
extern "C" int mhip_conv_f32_patch_geom(int out_c, int in_c, int kh, int kw, int stride, int pad, int in_h, int in_w, int out_h, int out_w, int *outv, int cap) {
    mhip_conv_f32_t p;
    shape_of(&p, out_c, in_c, kh, kw, stride, pad, in_h, in_w, out_h, out_w);
    fpatch_geom_t g;
    if (out_c <= 0 || !fpatch_geom(&p, &g, 1)) return 0;
    const int n = 30;
    if (cap < n) return 0;
    memcpy(outv, &g, n * sizeof(int));
    return n;
}

static unsigned long g_patch_launches = 0;
extern "C" unsigned long mhip_conv_f32_patch_launches(void) { return g_patch_launches; }

template <int BM, int WM, int WN>
static int launch_patch(const mhip_conv_f32_t *p, fpatch_geom_t &g) {
    auto kern = conv_f32_patch<BM, WM, WN>;
    static int cus = 0;
    if (!cus) {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return mhip_check(hipErrorUnknown, "conv_f32_patch attribute");
        cus = prop.multiProcessorCount;
    }
    const unsigned noc = (unsigned)((p->out_c + BM - 1) / BM);
    if (noc > 65535u) return -2;
    int slots = 0; // tests: "persist_slots" forces few workgroups, so that small inputs exercise long runs of tiles
    mhip_conv_i8_tune_get("persist_slots", &slots);
    unsigned gx = (unsigned)(slots > 0 ? slots : cus) / noc; // one 8-wave workgroup per CU (registers: two waves per SIMD)
    if (gx < 1) gx = 1;
    if (gx > g.ntiles) gx = g.ntiles;
    const unsigned per = (g.ntiles + gx - 1) / gx;
    gx = (g.ntiles + per - 1) / per;
    g.per = per;
    const size_t tabb = ((size_t)g.tab_ints * 4 + 255) & ~(size_t)255;
    hipLaunchKernelGGL(kern, dim3(gx, noc), dim3(P_NT), (size_t)g.lds_bytes, mhip_stream_native(), *p, g, (const int *)p->w_patch,
                       (const int8_t *)p->w_patch + tabb);
    g_patch_launches++;
    return mhip_check(hipGetLastError(), "conv_f32_patch");
}

// -2: not a shape this kernel takes (the caller goes on to conv_f32_split), else the launch result
int conv_f32_try_patch(const mhip_conv_f32_t *p) {
    if (!p->w_patch || p->use_mfma != 3) return -2;
    fpatch_geom_t g;
    if (!fpatch_geom(p, &g, p->frames)) return -2;
    if (p->add && p->add_stride != p->out_stride) return -2;
    if (g.BM == 128) return launch_patch<128, 2, 4>(p, g);
    if (g.BM == 64) return launch_patch<64, 1, 8>(p, g);
    return launch_patch<32, 1, 8>(p, g);
}
