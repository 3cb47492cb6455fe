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
//     a pixel's record is 32 bytes = [8 x hi | 8 x mid].  The patch ring holds two chunks;
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

#define P_NT 512   // threads per workgroup
#define P_NB 2     // patch ring slots (chunks)
#define P_PRCAP 96 // patch rows, at most
#define P_MAGIC 0x35504650

#ifdef FPATCH_STAMPS // diagnostic build (tools/stamps_build.sh fpatch; read with tools/fpatch_stamps.py): where a K step goes
__device__ unsigned long long fpatch_stamp_sums[8];
extern "C" int mhip_fpatch_stamps(unsigned long long *out, int reset) {
    unsigned long long z[8] = {0};
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(fpatch_stamp_sums), sizeof z) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(fpatch_stamp_sums), z, sizeof z) != hipSuccess) return -1;
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
#ifndef FPATCH_PRIO // experiment: 1 = the staging phase runs at raised wave priority (its few instructions ahead of the mate's MFMAs)
#define FPATCH_PRIO 0
#endif
#ifndef FPATCH_ABL // timing-only ablations (tools/stamps_build.sh fpabl N; wrong results): 1 no MFMAs, 2 no split / LDS writes of the
#define FPATCH_ABL 0 // patch, 4 no patch loads, 8 no fragment reads; conv_f32_prec: 16 no DMA waits, 32 no patch DMA
#endif

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
    int ngrp, nitems;          // 4-column groups per patch row, fetch items per chunk (threads that fetch) = PR * ngrp * (8 / cpi)
    int BM, kp, oc_pad;        // channel tile, weight row length (bf16 elements), weight rows per plane
    int tab_ints;              // ints of the table block in front of the weight planes
    int ndummy;                // dummy units at the end of a tile's stream (table entry -1: multiply zeros)
    int cpi;                   // channels per fetch item (8 | 4 | 2)
    int bn;                    // pixels per tile (256 | 512)
    int woff, poff, lds_bytes; // LDS byte offsets of the weight stages and the patch ring, total
    int nb, rec, ndma;         // patch ring slots (2; 2 | 4 for record input); input in record format; 1 KB LDS-DMA blocks per slot
    unsigned total_pix, ntiles, nsegs, in_bytes, per, out_bytes;
    pdiv_t dSW, dHo, dHV, dNS, dgrp, dPWP;
};

__device__ __forceinline__ int pa_lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 2)) << 4); }
__device__ __forceinline__ float psilu_fast(float v) { // as conv_f32_split's three-product mode: v_exp_f32 / v_rcp_f32
    const float e = __builtin_amdgcn_exp2f(v * -1.44269504088896341f);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}

// N floats (one pixel, N consecutive channels) -> N / 2 dwords of hi, N / 2 of mid.  hi = bf16(x) (round to nearest even), mid = bf16(x - hi);
// the subtraction is exact.  Not finite (hi = +-inf or NaN, i.e. |x| >= 2^128 - 2^119 or x not finite): mid = 0, so that the product sums
// see the inf / NaN once instead of the NaN an inf - inf residual would be
template <int N>
__device__ __forceinline__ void psplit(const float (&x)[N], int (&hi)[N / 2], int (&mid)[N / 2]) {
#pragma unroll
    for (int i = 0; i < N / 2; i++) {
        const f32x2 v = {x[2 * i], x[2 * i + 1]};
        const int h = __builtin_bit_cast(int, __builtin_convertvector(v, bf16x2));
        const float h0 = __int_as_float(h << 16), h1 = __int_as_float(h & (int)0xffff0000);
        f32x2 r; // two plain subtractions (left to itself the compiler packs them into a v_pk_add_f32 behind two register moves)
        asm("v_sub_f32 %0, %1, %2" : "=v"(r[0]) : "v"(v[0]), "v"(h0));
        asm("v_sub_f32 %0, %1, %2" : "=v"(r[1]) : "v"(v[1]), "v"(h1));
        r[0] = __builtin_isfinite(h0) ? r[0] : 0.0f;
        r[1] = __builtin_isfinite(h1) ? r[1] : 0.0f;
        hi[i] = h;
        mid[i] = __builtin_bit_cast(int, __builtin_convertvector(r, bf16x2));
    }
}

// BM = output channels per workgroup; waves WM (channels) x WN (pixels), WM * WN == 8; CPI = channels of a chunk per fetch item
// (8 | 4 | 2: the fewer, the more threads share a chunk's loads, split and LDS writes -- the host picks the smallest that keeps the
// items within the workgroup's 512 threads)
// P_BN = pixels per tile: 256, or 512 under the 64- / 32-channel tiles (a wave then has the 48 / 24 MFMAs per step that make a
// barrier interval worth its fixed cost; with 256 pixels those layers ran 24 / 12 per step and were bound by the step skeleton)
// RECIN > 0 (mhip_conv_f32_t.in_rec where conv_f32_prec's four ring slots do not fit: the large stride-2 patches): the input arrives in record
// format, so a chunk's staging is a plain COPY -- RECIN 16-byte half records per thread through registers into the slot (same LDS layout,
// same schedule, same two slots), no split: ~12 instead of ~140 vector instructions per chunk and thread.  (CPI is unused then.)
template <int BM, int WM, int WN, int CPI, bool DUMMY, int P_BN, int RECIN = 0>
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
    const int zrec = g.woff - 2 * P_PRCAP * 8 - 64 - g.poff;  // a zero pixel record (relative to the patch ring): what dummy units read
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
    if (tid < 16) ((int *)(patch + zrec))[tid] = 0;

    // ---- this thread's fetch item: patch row `ir`, columns 4 * igq .. + 3, channels ich * CPI .. + CPI - 1 of a chunk
    constexpr int NSUB = 8 / CPI; // items per (row, column group)
    const bool has_item = tid < g.nitems;
    const int icell = tid / NSUB, ich = tid % NSUB;
    const int ir = (int)pdiv((unsigned)icell, g.dgrp), igq = icell - ir * g.ngrp;
    // A pixel's record in the ring: 32 bytes = [8 x hi | 8 x mid] at pixel * 32, NO bank swizzle: the reader's address is then one
    // add per fragment (patch pixel of the lane's row + the unit's offset, both pre-multiplied), the mid half an immediate offset;
    // the two lane groups of a 16-byte read that meet in a bank (pixels 8 apart) cost LDS cycles the staging phase has, where the
    // ~30 vector instructions per step the swizzle cost were what bounded it (the staging wave issues at half rate beside its
    // mate's MFMAs: profiles/r05_experiments.md)
    int8_t *aitem[4]; // LDS address (slot 0) of this item's share of the records of its four columns
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int v = 4 * igq + i;
        aitem[i] = patch + (ir * g.PWP + (g.s == 2 ? (v >> 1) + (v & 1) * g.PWH : v)) * 32 + ich * (CPI * 2);
    }
    const int slot_bytes = g.slotpix * 32;
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
            rowtab[(t & 1) * P_PRCAP + tid] = make_int2(ok ? (int)(f * (unsigned)p.in_stride + (unsigned)(iy * g.W_in) * (RECIN ? 32u : 4u)) : -1,
                                                        (int)st * g.SW * g.s - g.pad - g.dx);
        }
    };
    // RECIN: item q of this thread = half record tid + 512 q of the slot (patch position / 2, hi | mid); its byte offset for the tile being
    // fetched lives in rvo[] (recomputed when the fetch stream moves on to the next tile, as in conv_f32_prec)
    constexpr int NRI = RECIN ? RECIN : 1;
    v4i rreg[NRI];
    unsigned rvo[NRI];
    auto tile_rvo = [&](unsigned t) __attribute__((always_inline)) {
        const int2 *rtab = rowtab + (t & 1u) * P_PRCAP;
#pragma unroll
        for (int q = 0; q < NRI; q++) {
            const unsigned it = (unsigned)(tid + q * P_NT), pos = it >> 1;
            const unsigned r = pdiv(pos, g.dPWP);
            const int cp = (int)(pos - r * (unsigned)g.PWP);
            const int v = g.s == 2 ? (cp < g.PWH ? 2 * cp : 2 * (cp - g.PWH) + 1) : cp;
            const bool in = (int)r < g.PR;
            const int2 rt = rtab[in ? r : 0u];
            const int x = rt.y + v;
            const bool ok = in && rt.x != -1 && x >= 0 && x < g.W_in;
            rvo[q] = ok ? (unsigned)rt.x + (unsigned)x * 32u + (it & 1u) * 16u : 0xffffffffu;
        }
    };
    v4i breg[CPI]; // the chunk in flight: channel ich * CPI + j, 4 columns
    const int irc = ir < g.PR ? ir : 0;
    int2 rt_cur = make_int2(-1, 0), rt_nxt = make_int2(-1, 0); // this thread's row-table entry for the tile being computed / the next one
    auto fetch_patch = [&](bool next_tile, int chunk) __attribute__((always_inline)) { // chunk of the current or the next tile
        if (RECIN) { // (the caller has moved rvo[] on to the next tile where the stream crossed over)
            const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)chunk * 8u * plane_bytes));
#pragma unroll
            for (int q = 0; q < NRI; q++) rreg[q] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(xrs, rvo[q], so, 0));
            return;
        }
        // no divergent branch here: a select per lane (threads without an item, rows / columns outside the image: an offset
        // the buffer unit's range check turns into zeros, no memory access), the chunk in the SCALAR offset; the row-table entry
        // comes from registers (an LDS round trip in front of every fetch was 150+ cycles of the staging phase)
        const int2 rt = next_tile ? rt_nxt : rt_cur;
        const int x = rt.y + 4 * igq;
        const bool ok = has_item && rt.x != -1 && x >= 0 && x < g.W_in;
        const unsigned vo = ok ? (unsigned)rt.x + (unsigned)x * 4u : 0xffffffffu;
        // (the item's channel group goes into the per-lane offset: it differs inside a wave)
        const unsigned vc = ok ? vo + (unsigned)(ich * CPI) * plane_bytes : 0xffffffffu;
        unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)chunk * 8u * plane_bytes));
#pragma unroll
        for (int j = 0; j < CPI; j++) {
            if (FPATCH_ABL & 4) breg[j] = (v4i){(int)vc, (int)so, j, 0};
            else breg[j] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(xrs, vc, so, 0));
            so += plane_bytes;
        }
    };
    auto commit_patch = [&](int slot) __attribute__((always_inline)) {
        if (RECIN) {
#pragma unroll
            for (int q = 0; q < NRI; q++)
                if ((tid + q * P_NT) * 16 < g.slotpix * 32) *(v4i *)(patch + slot * (g.slotpix * 32) + (tid + q * P_NT) * 16) = rreg[q];
            return;
        }
        if (FPATCH_ABL & 2) {
#pragma unroll
            for (int j = 0; j < CPI; j++) asm volatile("" ::"v"(breg[j]));
        } else if (has_item) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float x[CPI];
#pragma unroll
                for (int j = 0; j < CPI; j++) x[j] = __int_as_float(breg[j][i]);
                int hi[CPI / 2], mid[CPI / 2];
                psplit<CPI>(x, hi, mid);
                int8_t *a = aitem[i] + slot * slot_bytes;
                if (CPI == 8) {
                    *(v4i *)a = (v4i){hi[0], hi[1 % (CPI / 2)], hi[2 % (CPI / 2)], hi[3 % (CPI / 2)]};
                    *(v4i *)(a + 16) = (v4i){mid[0], mid[1 % (CPI / 2)], mid[2 % (CPI / 2)], mid[3 % (CPI / 2)]};
                } else if (CPI == 4) {
                    *(int2 *)a = make_int2(hi[0], hi[1 % (CPI / 2)]);
                    *(int2 *)(a + 16) = make_int2(mid[0], mid[1 % (CPI / 2)]);
                } else {
                    *(int *)a = hi[0];
                    *(int *)(a + 16) = mid[0];
                }
            }
        }
    };

    // ---- weights: row oc0 + tid / ATPR of both planes, elements (tid % ATPR) * AE .. of the step (as conv_f32_split)
    const int arow = tid / ATPR, akc = (tid % ATPR) * AE;
    // (buffer loads: a 32-bit per-lane offset, the (plane, step) part in the scalar offset: no 64-bit row pointer and address arithmetic per load)
    const unsigned wplane = (unsigned)g.oc_pad * (unsigned)g.kp * 2u;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)wpl, 0, (int)(2u * wplane), 0x00020000);
    const unsigned wvoff = ((unsigned)(oc0 + arow) * (unsigned)g.kp + (unsigned)akc) * 2u;
    int aregs[2][2][AD];
    auto fetch_w = [&](int ks, int (&areg)[2][AD]) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            const unsigned so = (unsigned)pl * wplane + (unsigned)ks * 64u;
            if (AE == 8) { const v4i t = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, so, 0)); areg[pl][0] = t[0]; areg[pl][1 % AD] = t[1]; areg[pl][2 % AD] = t[2]; areg[pl][3 % AD] = t[3]; }
            else if (AE == 4) { const auto t = __builtin_amdgcn_raw_buffer_load_b64(wrs, wvoff, so, 0); areg[pl][0] = (int)t[0]; areg[pl][1 % AD] = (int)t[1]; }
            else areg[pl][0] = (int)__builtin_amdgcn_raw_buffer_load_b32(wrs, wvoff, so, 0);
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
    const int8_t *pbase[NI]; // LDS address of the record of tap (0, 0), slot 0, of the lane's A rows
    unsigned ooff[NI]; // byte offset (frame + position inside a channel plane) of the lane's 4 result pixels, ~0 = none
    auto tile_setup = [&](unsigned t) __attribute__((always_inline)) {
        const int V0 = tile_v0(t);
#pragma unroll
        for (int n = 0; n < NI; n++) {
            const unsigned q = t * P_BN + (unsigned)(wn * TN + n * 16 + fr); // A operand: pixel fr of MFMA tile n
            const unsigned R = pdiv(q, g.dSW), xs = q - R * (unsigned)g.SW;
            const unsigned seg = pdiv(R, g.dHo), y = R - seg * (unsigned)g.H_out;
            const int prow = (int)(seg * (unsigned)g.HV + y * (unsigned)g.s) - V0;
            pbase[n] = patch + (q < g.total_pix ? prow * g.PWP + (int)xs : 0) * 32;
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

#ifdef FPATCH_STAMPS
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_last = 0, st_steps = 0;
#endif
    // A K step is two PHASES with a barrier behind each, and the two waves of a SIMD (w, w + 4) run ONE PHASE APART (waves 4-7 take
    // one extra barrier before the loop, waves 0-3 one after it): while one multiplies, its mate stages.
    //   R(t): [the chunk commit the schedule names] -> weights of step t + 1 (registers -> the other stage) -> the fragments of step t
    //         (unit table, patch, this stage's weights) into registers -> load issue (weights of step t + 3, the next chunk)
    //   M(t): the step's 48 MFMAs, nothing else.
    // First round 5 form: one barrier per step, every wave in the same phase: stamps put a step at body 1630 (the SIMD's 1536 matrix
    // cycles) + commit 410 + load issue 370 + barrier 790 + rest 310 = 3500 cycles -- the matrix pipe idle half the time.
    // LDS hazards under the offset: everything a phase R(t) WRITES (patch slot, stage (t + 1) & 1) was last read in an R(t - 1), which
    // for either group ended at the barrier before this R(t) begins; everything it READS was written in an R(t - 1) or earlier.
    bf16x8 xh[NI], xm[NI], wh[MI], wmid[MI];
    auto read_frags = [&](int t, int buf, const int e) __attribute__((always_inline)) {
        const int8_t *ap = wst + buf * WSTAGE;
#pragma unroll
        for (int c = 0; c < NI; c++) {
            const int8_t *a = pbase[c] + e; // e = the unit's offset in BYTES (slot included)
            // dummy units (the stream padded to an even number of steps; table entry -1) read a ZERO record: their weights are zero,
            // but 0 x inf would be a NaN the reference does not have (the float twins do hold infs)
            if (DUMMY) a = e < 0 ? patch + zrec : a;
            if (FPATCH_ABL & 8) { xh[c] = __builtin_bit_cast(bf16x8, (v4i){(int)(size_t)a, c, t, lane}); xm[c] = xh[c]; continue; }
            xh[c] = __builtin_bit_cast(bf16x8, *(const v4i *)a);
            xm[c] = __builtin_bit_cast(bf16x8, *(const v4i *)(a + 16));
        }
#pragma unroll
        for (int a = 0; a < MI; a++) {
            const int o = pa_lds_off(wm * TM + a * 16 + fr, fc);
            if (FPATCH_ABL & 8) { wh[a] = __builtin_bit_cast(bf16x8, (v4i){o, a, t, lane}); wmid[a] = wh[a]; continue; }
            wh[a] = __builtin_bit_cast(bf16x8, *(const v4i *)(ap + o));
            wmid[a] = __builtin_bit_cast(bf16x8, *(const v4i *)(ap + APLANE + o));
        }
    };
    auto phase_m = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < MI; a++)
#pragma unroll
            for (int c = 0; c < NI; c++) { // pixels are the A operand: a lane ends with 4 consecutive pixels of one channel
                if (FPATCH_ABL & 1) { asm volatile("" ::"v"(xm[c]), "v"(xh[c]), "v"(wh[a]), "v"(wmid[a])); continue; }
                acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm[c], wh[a], acc[a][c], 0, 0, 0);   // mid * hi
                acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[c], wmid[a], acc[a][c], 0, 0, 0); // hi * mid
                acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[c], wh[a], acc[a][c], 0, 0, 0);   // hi * hi
            }
    };
    auto barrier_lds = [&]() __attribute__((always_inline)) { // this wave's LDS operations done (loads in flight stay in flight), then the barrier
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    // top of R(t): commit the chunk the schedule names (its loads were issued >= 2 steps ago); the next chunk's loads are issued AFTER
    // the step's weight loads (vmcnt is in order: the weights, consumed one step later, must not queue behind them).  The schedule
    // word and the unit offset of step t were read from LDS during R(t - 1) (tab_sc / tab_e): no LDS round trip in front of either
    auto patch_commit = [&](int sc) __attribute__((always_inline)) {
        if (sc) commit_patch((sc - 1) & (P_NB - 1));
    };
    unsigned tcur = 0; // the tile being computed (RECIN: patch_fetch moves the offsets on from it)
    auto patch_fetch = [&](int sc) __attribute__((always_inline)) {
        if (sc) {
            const int nx = sc; // chunk after the one just committed: of this tile, or nchunk + chunk of the next
            const bool nxt = nx >= g.nchunk;
            if (RECIN && nx == g.nchunk) tile_rvo(tcur + 1); // (the next tile's row table: written in this tile's R(0); no fetch for it before step 2)
            fetch_patch(nxt, nxt ? nx - g.nchunk : nx);
        }
    };

    // ---- prologue: tables, the first two chunks of the first tile, the first weight steps
    // this workgroup's run of tiles: an even split of the tile list over the grid's x extent
    const unsigned t_first = (unsigned)(((unsigned long long)blockIdx.x * g.ntiles) / gridDim.x);
    const unsigned t_end = (unsigned)(((unsigned long long)(blockIdx.x + 1) * g.ntiles) / gridDim.x);
    fill_rowtab(t_first);
    __syncthreads();
    rt_cur = rowtab[(t_first & 1) * P_PRCAP + irc];
    if (RECIN) tile_rvo(t_first);
    // the state every later tile starts in: what the PREVIOUS tile's steps would have done for this tile's first chunks (the
    // schedule entries that name chunk nchunk + c), in order -- chunk 0's loads, then per entry: commit, fetch the next
    fetch_patch(false, 0);
    for (int i = 0; i < g.nsteps; i++) {
        const int sc = __builtin_amdgcn_readfirstlane(sched[i]);
        if (sc > g.nchunk) {
            const int cc = sc - 1 - g.nchunk;
            commit_patch(cc & (P_NB - 1));
            fetch_patch(false, cc + 1);
        }
    }
    fetch_w(0, aregs[0]);
    commit_w(0, aregs[0]);
    fetch_w(1, aregs[1]);
    fetch_w(2, aregs[0]);
    __syncthreads();

    const int nsteps = g.nsteps;
    const bool late = wv >= 4; // the second wave of its SIMD: one phase behind
    int tab_e = dutab[fc], tab_sc = sched[0];
    if (late) __builtin_amdgcn_s_barrier();
#ifdef FPATCH_STAMPS
    st_last = __builtin_readcyclecounter();
#endif
    for (unsigned t = t_first; t < t_end; t++) {
        tcur = t;
#ifdef FPATCH_STAMPS
        st_steps += (unsigned long long)nsteps;
#endif
        tile_setup(t);
        for (int ks = 0; ks < nsteps; ks += 2) {
            int kq = ks + 3; // weights are fetched three steps ahead (two register sets, two stages); they wrap into the next tile
            if (kq >= nsteps) kq -= nsteps;
            int kq1 = ks + 4;
            if (kq1 >= nsteps) kq1 -= nsteps;
            const int k2 = ks + 2 < nsteps ? ks + 2 : 0;
            if (ks == 0) fill_rowtab(t + 1); // the other table held tile t - 1's rows: its last chunk was fetched during tile t - 1
            STAMP(4); // loop overhead; tile setup and epilogue when a tile began
            // ---- R(ks)
            if (FPATCH_PRIO) __builtin_amdgcn_s_setprio(1);
            {
                const int e = tab_e, sc = __builtin_amdgcn_readfirstlane(tab_sc);
                tab_e = dutab[(ks + 1) * 4 + fc]; // for R(ks + 1): in registers by this phase's barrier
                tab_sc = sched[ks + 1];
                patch_commit(sc);
                STAMP(0); // the chunk commit (wait for its loads, split, LDS writes) when the schedule names one
                commit_w(1, aregs[1]);
                read_frags(ks, 0, e);
                fetch_w(kq, aregs[1]);
                patch_fetch(sc);
            }
            STAMP(1); // staging: next weights into LDS, fragment reads, load issue
            if (FPATCH_PRIO) __builtin_amdgcn_s_setprio(0);
            barrier_lds();
            STAMP(3);
            phase_m();
            __builtin_amdgcn_sched_barrier(0);
            STAMP(2); // the MFMAs
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            STAMP(3); // barriers
            // ---- R(ks + 1)
            if (FPATCH_PRIO) __builtin_amdgcn_s_setprio(1);
            {
                const int e = tab_e, sc = __builtin_amdgcn_readfirstlane(tab_sc);
                tab_e = dutab[k2 * 4 + fc];
                tab_sc = sched[k2];
                // the next tile's row-table entry: written in an R(0) (waves 0 / 1), two barriers before any wave's R(1); the host's
                // schedule has no fetch for the next tile before step 2
                if (ks == 0) rt_nxt = rowtab[((t + 1) & 1) * P_PRCAP + irc];
                patch_commit(sc);
                STAMP(0);
                commit_w(0, aregs[0]);
                read_frags(ks + 1, 1, e);
                fetch_w(kq1, aregs[0]);
                patch_fetch(sc);
            }
            STAMP(1);
            if (FPATCH_PRIO) __builtin_amdgcn_s_setprio(0);
            barrier_lds();
            STAMP(3);
            phase_m();
            __builtin_amdgcn_sched_barrier(0);
            STAMP(2);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            STAMP(3);
        }
        rt_cur = rt_nxt;
        // ---- store: a lane holds 4 consecutive pixels (one 16-byte store) of channel row fr of every MFMA tile.  The operand of a
        // fused residual Add is loaded for ALL of the lane's results first (the fragment registers are dead here), then added:
        // one load, one wait, one add, one store per result in turn was a memory latency per result (0.3 ms on the 80 x 80 layers)
        if (p.add) {
            v4f addv[MI][NI];
#pragma unroll
            for (int c = 0; c < NI; c++)
#pragma unroll
                for (int a = 0; a < MI; a++) {
                    const int oc = oc0 + wm * TM + a * 16 + fr;
                    const bool ok = ooff[c] != 0xffffffffu && oc < p.out_c;
                    const size_t o = ok ? (size_t)ooff[c] + (size_t)oc * hw * 4u : 0; // (add_stride == out_stride: checked by the launcher)
                    addv[a][c] = *(const v4f *)((const char *)p.add + o);
                }
#pragma unroll
            for (int c = 0; c < NI; c++)
#pragma unroll
                for (int a = 0; a < MI; a++) {
                    const int oc = oc0 + wm * TM + a * 16 + fr;
                    if (ooff[c] != 0xffffffffu && oc < p.out_c) {
                        v4f r = acc[a][c];
                        if (p.silu) {
#pragma unroll
                            for (int j = 0; j < 4; j++) r[j] = psilu_fast(r[j]);
                        }
                        *(v4f *)((char *)p.out + (size_t)ooff[c] + (size_t)oc * hw * 4u) = r + addv[a][c];
                    }
                }
        } else {
#pragma unroll
            for (int c = 0; c < NI; c++)
#pragma unroll
                for (int a = 0; a < MI; a++) {
                    const int oc = oc0 + wm * TM + a * 16 + fr;
                    if (ooff[c] != 0xffffffffu && oc < p.out_c) {
                        v4f r = acc[a][c];
                        if (p.silu) {
#pragma unroll
                            for (int j = 0; j < 4; j++) r[j] = psilu_fast(r[j]);
                        }
                        *(v4f *)((char *)p.out + (size_t)ooff[c] + (size_t)oc * hw * 4u) = r;
                    }
                }
        }
#pragma unroll
        for (int c = 0; c < NI; c++)
#pragma unroll
            for (int a = 0; a < MI; a++) acc[a][c] = bias4[a];
    }
    if (!late) __builtin_amdgcn_s_barrier(); // pairs with the late waves' last one
#ifdef FPATCH_STAMPS
    if (lane == 0) {
        for (int i = 0; i < 5; i++) atomicAdd(&fpatch_stamp_sums[i], st_acc[i]);
        atomicAdd(&fpatch_stamp_sums[5], st_steps);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// conv_f32_prec: the same convolution when the INPUT ARRIVES AS RECORDS (mhip_conv_f32_t.in_rec): the producing convolution
// (conv_f32_split / conv_f32_stem with out_rec) has already cut its results into the two bf16 pieces and written them channels-last,
// [in_c / 8 chunks][H][W] records of 32 bytes = [8 x hi | 8 x mid] -- exactly what conv_f32_patch's staging phase builds in LDS.  The
// planner uses the format for a tensor whose only reader is one such convolution (a C3 bottleneck's 1 x 1 -> 3 x 3; mars_plan.c rec_pairs).
// What that removes from a K step's staging phase (profiles/r05_experiments.md: ~75 + ~130 vector instructions issued at half rate
// beside the SIMD mate's MFMAs bound conv_f32_patch): the patch loads into registers, the split, the LDS writes -- a chunk's slot is
// filled by LDS-DMA (buffer_load ... lds, 1 KB per wave instruction, zero fill outside the image by the buffer's range check), issued by
// inline assembly the compiler neither counts nor guards, into a ring of FOUR slots (a chunk is requested two chunk periods before its
// first reader).  The waits for it are counted by hand: a wave knows how many vector-memory instructions it has issued after the DMA it
// needs (they retire in order), kept as scalar "ages" of the pending chunks (yp*, a FIFO).  Weights: through registers, as conv_f32_patch.
// Phases, barriers, MFMA order, tile walk: conv_f32_patch's; the epilogue is branch-free (buffer loads / stores).
// Same arithmetic as conv_f32_patch on the same values: the two forms' outputs are bit-identical (tools/layer_time.py --chain).
// Measured (batch 256, the 3 x 3 of a bottleneck pair, us): 160 x 160 x 32: 1044 -> 851; 80 x 80 x 64: 614 -> 537; 40 x 40 x 128: 426 -> 388;
// 20 x 20 x 256: 429 -> 380.  Not for the patches that leave room for two slots only (the large stride-2 layers): slower there.
// s_waitcnt vmcnt(n) for a wave-uniform run-time n, rounded DOWN to one of three immediates (any smaller count is correct too, only
// slower): HI = everything but what one step issues at most (its chunk DMA and two weight loads) may stay in flight -- the usual case, a
// chunk is needed steps after its DMA; 2 = the chunk was issued in this very step, only the weight loads behind it; else everything.
// (A binary tree down to the exact immediate cost ~450 cycles per wait: twelve scalar branches.)
template <int HI>
__device__ __forceinline__ void pwait_vm(int n) {
    if (n >= HI) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HI) : "memory");
    else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// One LDS-DMA instruction: lane l's 16 bytes at buffer offset voff + soff land at LDS byte lds_wave_base + 16 l (zeros when the offset is
// outside the buffer).  Inline assembly, not the builtin: the compiler then neither counts it (every wait for these is hand-counted
// anyway) nor guards its address register -- with the builtin it put an s_waitcnt vmcnt(0) in front of the next write to the VGPR that
// had held a DMA's offset, i.e. once per K step.  M0 = the LDS base: nothing else in this kernel uses it.
__device__ __forceinline__ void pdma16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, const void *lds_wave_base) {
    const unsigned l = (unsigned)(size_t)(const __attribute__((address_space(3))) void *)lds_wave_base;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(l), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

// NDW = LDS-DMA instructions per wave and chunk (ceil(slot KB / 8), rounded up to 2 | 3 | 4 | 6 | 8)
template <int BM, int WM, int WN, bool DUMMY, int P_BN, int NDW>
__global__ __launch_bounds__(P_NT, 2) void conv_f32_prec(const mhip_conv_f32_t p, const fpatch_geom_t g, const int *__restrict__ tabs,
                                                         const int8_t *__restrict__ wpl) {
    constexpr int TM = BM / WM, TN = P_BN / WN;
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int APLANE = BM * 64, WSTAGE = 2 * APLANE;
    constexpr int AE = BM * 32 / P_NT;     // weight elements per thread, plane and step: 8 | 4 | 2
    constexpr int ATPR = 32 / AE, AD = AE / 2;
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    int *dutab = (int *)lds;
    int *sched = dutab + g.nsteps * 4; // low half: 1 + chunk whose DMA this step issues; bit 16: a chunk is first read in the next step
    int2 *rowtab = (int2 *)(lds + g.woff - 2 * P_PRCAP * 8);
    const int zrec = g.woff - 2 * P_PRCAP * 8 - 64 - g.poff;
    int8_t *wst = lds + g.woff;
    int8_t *patch = lds + g.poff;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv % WM, wn = wv / WM;
    const int fr = lane & 15, fc = lane >> 4;
    const int oc0 = (int)blockIdx.y * BM;
    const unsigned hw = (unsigned)(g.H_out * g.W_out);
    const unsigned chunk_bytes = (unsigned)(g.H_in * g.W_in) * 32u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, (int)g.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, 0, (int)g.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void *)(p.add ? (const void *)p.add : (const void *)p.out), 0, (int)g.out_bytes, 0x00020000);

    for (int i = tid; i < g.nsteps * 5; i += P_NT) dutab[i] = tabs[i];
    if (tid < 16) ((int *)(patch + zrec))[tid] = 0;

    // ---- weights: through registers, as conv_f32_patch (row oc0 + tid / ATPR of both planes, elements (tid % ATPR) * AE .. of the step; two
    // register sets, two LDS stages, fetched three steps ahead).  By LDS-DMA too (the first form of this kernel, four stages): a
    // 128-channel stage is 16 KB per step and the DMA path fills LDS at ~27 B/clk/CU (probe, round 4) -- 600 cycles of it per step beside
    // the patch's, every wait for either grew long (stamps: patch issue 620, DMA waits 510, barriers 1270 cycles per step on D40)
    const int arow = tid / ATPR, akc = (tid % ATPR) * AE;
    // (buffer loads: a 32-bit per-lane offset and the (plane, step) part in the scalar offset -- a 64-bit row pointer and its per-load address
    // arithmetic cost the registers whose spill put a reload, and with it an s_waitcnt vmcnt(0), in front of the K loop)
    const unsigned wplane_b = (unsigned)g.oc_pad * (unsigned)g.kp * 2u;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)wpl, 0, (int)(2u * wplane_b), 0x00020000);
    const unsigned wvoff = ((unsigned)(oc0 + arow) * (unsigned)g.kp + (unsigned)akc) * 2u;
    int aregs[2][2][AD];
    auto fetch_w = [&](int ks, int (&areg)[2][AD]) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            const unsigned so = (unsigned)pl * wplane_b + (unsigned)ks * 64u;
            if (AE == 8) { const v4i t = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, so, 0)); areg[pl][0] = t[0]; areg[pl][1 % AD] = t[1]; areg[pl][2 % AD] = t[2]; areg[pl][3 % AD] = t[3]; }
            else if (AE == 4) { const auto t = __builtin_amdgcn_raw_buffer_load_b64(wrs, wvoff, so, 0); areg[pl][0] = (int)t[0]; areg[pl][1 % AD] = (int)t[1]; }
            else areg[pl][0] = (int)__builtin_amdgcn_raw_buffer_load_b32(wrs, wvoff, so, 0);
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

    // ---- patch: instruction j of this wave fills block j * 8 + wv of a slot = patch positions (j * 8 + wv) * 32 + lane / 2, the lane's
    // half record (lane & 1: hi | mid).  The lane's byte offsets for the tile whose chunks are being issued live in registers (vo[]):
    // recomputed -- position -> (patch row, column), the row's offset from the tile's row table -- when the issue stream moves on to
    // the next tile, once per tile.  (Read from the row table at every issue, the LDS round trip beside the other waves' fragment
    // reads cost ~1500 cycles per issue: stamps.)
    const int ndw = g.ndma > wv ? (g.ndma - wv + 7) >> 3 : 0;
    auto tile_v0 = [&](unsigned t) __attribute__((always_inline)) {
        const unsigned R0 = pdiv(t * P_BN, g.dSW), seg0 = pdiv(R0, g.dHo);
        return (int)(seg0 * (unsigned)g.HV + (R0 - seg0 * (unsigned)g.H_out) * (unsigned)g.s);
    };
    auto fill_rowtab = [&](unsigned t) __attribute__((always_inline)) {
        if (tid < g.PR) {
            const unsigned V = (unsigned)tile_v0(t) + (unsigned)tid;
            const unsigned seg = pdiv(V, g.dHV), f = pdiv(seg, g.dNS), st = seg - f * (unsigned)g.nstrips;
            const int iy = (int)(V - seg * (unsigned)g.HV) - g.pad;
            const bool ok = t < g.ntiles && seg < g.nsegs && iy >= 0 && iy < g.H_in;
            rowtab[(t & 1) * P_PRCAP + tid] = make_int2(ok ? (int)(f * (unsigned)p.in_stride + (unsigned)(iy * g.W_in) * 32u) : -1,
                                                        (int)st * g.SW * g.s - g.pad - g.dx);
        }
    };
    unsigned vo[NDW];
    auto tile_vo = [&](unsigned t) __attribute__((always_inline)) {
        const int2 *rtab = rowtab + (t & 1u) * P_PRCAP;
#pragma unroll
        for (int j = 0; j < NDW; j++)
            if (j < ndw) {
                const unsigned pos = (unsigned)((j * 8 + wv) * 32 + (lane >> 1));
                const unsigned r = pdiv(pos, g.dPWP);
                const int cp = (int)(pos - r * (unsigned)g.PWP);
                const int v = g.s == 2 ? (cp < g.PWH ? 2 * cp : 2 * (cp - g.PWH) + 1) : cp;
                const bool in = (int)r < g.PR;
                const int2 rt = rtab[in ? r : 0u];
                const int x = rt.y + v;
                const bool ok = in && rt.x != -1 && x >= 0 && x < g.W_in;
                vo[j] = ok ? (unsigned)rt.x + (unsigned)x * 32u + (unsigned)(lane & 1) * 16u : 0xffffffffu;
            }
    };
    // chunk `rel` of tile t (rel >= nchunk: chunk rel - nchunk of tile t + 1)
    const int slot_bytes = g.slotpix * 32;
    auto issue_patch = [&](unsigned t, int rel) __attribute__((always_inline)) {
        const bool nxt = rel >= g.nchunk;
        const int c = nxt ? rel - g.nchunk : rel;
        if (rel == g.nchunk) tile_vo(t + 1); // the stream moves on to the next tile (its row table: written in an R(0); the schedule has no DMA for it before step 2)
        int8_t *dst = patch + (c & (g.nb - 1)) * slot_bytes + wv * 1024;
        const unsigned so = (unsigned)c * chunk_bytes;
#pragma unroll
        for (int j = 0; j < NDW; j++)
            if (j < ndw && !(FPATCH_ABL & 32)) pdma16(xrs, vo[j], so, dst + j * 8192);
    };

    // ---- the compute side (as conv_f32_patch)
    // (registers: the 128-channel instantiation has none to spare -- a value that lives through the K loop unused (the output offsets, the
    // bias) is recomputed / re-read where it is needed; a spilled register's reload in front of the K loop makes the compiler wait
    // vmcnt(0) -- i.e. for every DMA in flight -- at the loop's first counted wait, in every step)
    const int8_t *pbase[NI];
    auto tile_setup = [&](unsigned t) __attribute__((always_inline)) {
        const int V0 = tile_v0(t);
#pragma unroll
        for (int n = 0; n < NI; n++) {
            const unsigned q = t * P_BN + (unsigned)(wn * TN + n * 16 + fr);
            const unsigned R = pdiv(q, g.dSW), xs = q - R * (unsigned)g.SW;
            const unsigned seg = pdiv(R, g.dHo), y = R - seg * (unsigned)g.H_out;
            const int prow = (int)(seg * (unsigned)g.HV + y * (unsigned)g.s) - V0;
            pbase[n] = patch + (q < g.total_pix ? prow * g.PWP + (int)xs : 0) * 32;
        }
    };
    auto out_off = [&](unsigned t, int n) __attribute__((always_inline)) { // byte offset (frame + position inside a channel plane) of the lane's 4 result pixels of MFMA tile n, ~0 = none
        const unsigned q4 = t * P_BN + (unsigned)(wn * TN + n * 16 + fc * 4);
        const unsigned R4 = pdiv(q4, g.dSW), xs4 = q4 - R4 * (unsigned)g.SW;
        const unsigned seg4 = pdiv(R4, g.dHo), y4 = R4 - seg4 * (unsigned)g.H_out;
        const unsigned f4 = pdiv(seg4, g.dNS), st4 = seg4 - f4 * (unsigned)g.nstrips;
        return q4 < g.total_pix ? f4 * (unsigned)p.out_stride + (y4 * (unsigned)g.W_out + st4 * (unsigned)g.SW + xs4) * 4u : 0xffffffffu;
    };
    v4f acc[MI][NI];
    float *sbias = (float *)(lds + g.woff - 2 * P_PRCAP * 8 - 64 - 512); // [BM] (the host leaves 512 bytes in front of the zero record)
    if (tid < BM) sbias[tid] = p.bias && oc0 + tid < p.out_c ? p.bias[oc0 + tid] : 0.f;
#pragma unroll
    for (int a = 0; a < MI; a++) {
        const int oc = oc0 + wm * TM + a * 16 + fr;
        const float b = p.bias && oc < p.out_c ? p.bias[oc] : 0.f;
#pragma unroll
        for (int c = 0; c < NI; c++) acc[a][c] = (v4f){b, b, b, b};
    }
    bf16x8 xh[NI], xm[NI], wh[MI], wmid[MI];
    auto read_frags = [&](int stage, const int e) __attribute__((always_inline)) {
        const int8_t *ap = wst + stage * WSTAGE;
#pragma unroll
        for (int c = 0; c < NI; c++) {
            const int8_t *a = pbase[c] + e;
            if (DUMMY) a = e < 0 ? patch + zrec : a;
            xh[c] = __builtin_bit_cast(bf16x8, *(const v4i *)a);
            xm[c] = __builtin_bit_cast(bf16x8, *(const v4i *)(a + 16));
        }
#pragma unroll
        for (int a = 0; a < MI; a++) {
            const int o = pa_lds_off(wm * TM + a * 16 + fr, fc);
            wh[a] = __builtin_bit_cast(bf16x8, *(const v4i *)(ap + o));
            wmid[a] = __builtin_bit_cast(bf16x8, *(const v4i *)(ap + APLANE + o));
        }
    };
    auto phase_m = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < MI; a++)
#pragma unroll
            for (int c = 0; c < NI; c++) {
                acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm[c], wh[a], acc[a][c], 0, 0, 0);   // mid * hi
                acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[c], wmid[a], acc[a][c], 0, 0, 0); // hi * mid
                acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[c], wh[a], acc[a][c], 0, 0, 0);   // hi * hi
            }
    };
    auto barrier_lds = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto barrier_raw = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: tables, the row table and DMA offsets of the first tile, what the previous tile's steps would have issued for
    // it (the schedule entries that name a chunk of the NEXT tile), the first three weight stages; everything landed
    // this workgroup's run of tiles: an even split of the tile list over the grid's x extent
    const unsigned t_first = (unsigned)(((unsigned long long)blockIdx.x * g.ntiles) / gridDim.x);
    const unsigned t_end = (unsigned)(((unsigned long long)(blockIdx.x + 1) * g.ntiles) / gridDim.x);
    fill_rowtab(t_first);
    __syncthreads();
    const int nsteps = g.nsteps;
    int npend = 0;
    tile_vo(t_first);
    for (int i = 0; i < nsteps; i++) {
        const int sc = __builtin_amdgcn_readfirstlane(sched[i]) & 0xffff;
        if (sc > g.nchunk) {
            issue_patch(t_first, sc - 1 - g.nchunk);
            npend++;
        }
    }
    npend--; // chunk 0 is read in step 0: waited for here, not by a step's wait
    fetch_w(0, aregs[0]);
    commit_w(0, aregs[0]);
    fetch_w(1, aregs[1]);
    fetch_w(2, aregs[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ages: vector-memory instructions this wave has issued after the DMA of each pending chunk (oldest first).  EVERY such instruction
    // counts, the compiler's too: two weight loads per step, the epilogue's loads and stores
    int yp0 = 0, yp1 = 0, yp2 = 0, yp3 = 0;
    const bool late = wv >= 4;
    int tab_e = dutab[fc], tab_w = sched[0];
    if (late) barrier_raw();
#ifdef FPATCH_STAMPS // (tools/stamps_build.sh fpatch; tools/fpatch_stamps.py --chain): 0 weights (LDS write, next loads), 1 patch DMA issue, 2 fragment reads, 3 MFMAs, 4 DMA waits, 5 barriers, 6 rest
    unsigned long long st_acc[7] = {0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_readcyclecounter(), st_steps = 0;
#endif
    // one staging phase R(ks) -- LDS stage BUF for its reads, the weight registers AREG (step ks + 1's, fetched two steps ago) -- and the
    // rest of the step.  Macros, not lambdas: the ages must stay scalar registers (captured by reference in a lambda they went to
    // scratch memory, as vector values)
#define PREC_PHASE_R(KS, BUF, AREG, K1, KQ)                                                                                            \
    {                                                                                                                                  \
        const int e = tab_e, w = __builtin_amdgcn_readfirstlane(tab_w), sc = w & 0xffff;                                               \
        tab_e = dutab[(K1) * 4 + fc];                                                                                                  \
        tab_w = sched[(K1)];                                                                                                           \
        STAMP(6);                                                                                                                      \
        commit_w((BUF) ^ 1, AREG);                                                                                                     \
        STAMP(0);                                                                                                                      \
        /* the chunk's DMA goes BEFORE this step's weight loads: the compiler's wait for those (vmcnt(2), a step later) then finds    \
           the DMA among the older instructions it completes anyway, not among the two it leaves in flight -- a full step to land.    \
           (The next tile's row table: written in an R(0); the schedule has no DMA for the next tile before step 2.) */                \
        if (sc) {                                                                                                                      \
            issue_patch(t, sc - 1);                                                                                                    \
            yp0 += ndw; yp1 += ndw; yp2 += ndw; yp3 += ndw;                                                                            \
            if (npend == 0) yp0 = 0;                                                                                                   \
            else if (npend == 1) yp1 = 0;                                                                                              \
            else if (npend == 2) yp2 = 0;                                                                                              \
            else yp3 = 0;                                                                                                              \
            npend++;                                                                                                                   \
        }                                                                                                                              \
        fetch_w((KQ), AREG);                                                                                                           \
        yp0 += 2; yp1 += 2; yp2 += 2; yp3 += 2;                                                                                        \
        STAMP(1);                                                                                                                      \
        read_frags((BUF), e);                                                                                                          \
        PREC_STAMP_LGKM;                                                                                                               \
        STAMP(2);                                                                                                                      \
        /* a chunk the NEXT step reads for the first time must have landed, in every wave, by the barrier in front of that step's     \
           first reader (waves 0-3's R(ks + 1)): this wave's share is waited for before that barrier -- the one behind this phase     \
           for waves 4-7, the one behind the MFMAs for waves 0-3 */                                                                    \
        int nwait = -1;                                                                                                                \
        if (w >> 16) {                                                                                                                 \
            nwait = yp0;                                                                                                               \
            yp0 = yp1; yp1 = yp2; yp2 = yp3;                                                                                           \
            npend--;                                                                                                                   \
        }                                                                                                                              \
        if (late && nwait >= 0 && !(FPATCH_ABL & 16)) pwait_vm<NDW + 2>(nwait);                                                                 \
        STAMP(4);                                                                                                                      \
        barrier_lds();                                                                                                                 \
        STAMP(5);                                                                                                                      \
        phase_m();                                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        STAMP(3);                                                                                                                      \
        if (!late && nwait >= 0 && !(FPATCH_ABL & 16)) pwait_vm<NDW + 2>(nwait);                                                                \
        STAMP(4);                                                                                                                      \
        barrier_raw();                                                                                                                 \
        STAMP(5);                                                                                                                      \
    }
#ifdef FPATCH_STAMPS
#define PREC_STAMP_LGKM asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define PREC_STAMP_LGKM do { } while (0)
#endif
    for (unsigned t = t_first; t < t_end; t++) {
#ifdef FPATCH_STAMPS
        st_steps += (unsigned long long)nsteps;
#endif
        tile_setup(t);
        for (int ks = 0; ks < nsteps; ks += 2) {
            int kq = ks + 3; // weights are fetched three steps ahead (two register sets, two stages); they wrap into the next tile
            if (kq >= nsteps) kq -= nsteps;
            int kq1 = ks + 4;
            if (kq1 >= nsteps) kq1 -= nsteps;
            const int k2 = ks + 2 < nsteps ? ks + 2 : 0;
            if (ks == 0) fill_rowtab(t + 1);
            PREC_PHASE_R(ks, 0, aregs[1], ks + 1, kq)
            PREC_PHASE_R(ks + 1, 1, aregs[0], k2, kq1)
        }
#undef PREC_PHASE_R
        // ---- store: a lane holds 4 consecutive pixels (16 bytes) of channel row fr of every MFMA tile.  Buffer loads / stores whose
        // offset is out of range for the lanes without a result: no branch, so every wave issues exactly MI * NI stores (and as many
        // loads of a fused Add's operand) per tile -- the ages stay exact -- and nothing the compiler loads is read behind a branch
        // (a loaded register whose only reader may be skipped stays "pending" for the compiler: it then put an s_waitcnt vmcnt(0) in
        // front of the K loop's first write to that register, in every step)
        {
            unsigned oo[NI];
#pragma unroll
            for (int c = 0; c < NI; c++) oo[c] = out_off(t, c);
            auto vso = [&](int a, int c) __attribute__((always_inline)) {
                const int oc = oc0 + wm * TM + a * 16 + fr;
                return oo[c] != 0xffffffffu && oc < p.out_c ? oo[c] + (unsigned)oc * hw * 4u : 0xffffffffu;
            };
            v4f addv[MI][NI];
            if (p.add) {
#pragma unroll
                for (int c = 0; c < NI; c++)
#pragma unroll
                    for (int a = 0; a < MI; a++) addv[a][c] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(ars, vso(a, c), 0, 0));
            }
#pragma unroll
            for (int c = 0; c < NI; c++)
#pragma unroll
                for (int a = 0; a < MI; a++) {
                    v4f r = acc[a][c];
                    if (p.silu) {
#pragma unroll
                        for (int j = 0; j < 4; j++) r[j] = psilu_fast(r[j]);
                    }
                    if (p.add) r += addv[a][c];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, r), ors, vso(a, c), 0, 0);
                    const float b = sbias[wm * TM + a * 16 + fr];
                    acc[a][c] = (v4f){b, b, b, b};
                    __builtin_amdgcn_sched_barrier(0); // one result at a time: interleaved, the SiLU temporaries of all of them cost 40 registers
                }
            const int nst = MI * NI * (p.add ? 2 : 1);
            yp0 += nst; yp1 += nst; yp2 += nst; yp3 += nst;
        }
    }
    if (!late) barrier_raw(); // pairs with the late waves' last one
#ifdef FPATCH_STAMPS
    if (lane == 0) {
        for (int i = 0; i < 7; i++) atomicAdd(&fpatch_stamp_sums[i], st_acc[i]);
        atomicAdd(&fpatch_stamp_sums[7], st_steps);
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // no LDS-DMA of this wave may land after the workgroup's LDS is handed on
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
// rec (record-format input, conv_f32_prec): an entry is the step whose staging phase ISSUES the chunk's LDS-DMA into its slot (no
// register staging: consecutive chunks may follow one step apart), NB = 2 | 4 slots; bit 16 of a step's word = "a chunk is read
// for the first time in the NEXT step" (the wave waits for it before it leaves this one).
static int fpatch_schedule(int nchunk, int U, int nsteps, int *sched /* [nsteps] */, int NB = P_NB, int rec = 0) {
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
        if (c < prev + (rec ? 1 : 2)) c = prev + (rec ? 1 : 2);
        if (c > first_read(gch) - 1) return 0;
        prev = c;
        if (c >= nsteps && c < 2L * nsteps) {
            const long rel = gch - nchunk; // chunk index relative to tile 1: may reach into tile 2 (>= nchunk)
            if (rel < 0 || rel >= 2L * nchunk || sched[c - nsteps]) return 0;
            sched[c - nsteps] = (int)rel + 1;
            if ((rec ? rel : rel + 1) >= nchunk && c - nsteps < 2) return 0; // the kernel reads the next tile's row table in R(1): no fetch for it before step 2
        }
    }
    if (rec)
        for (int c = 0; c < nchunk; c++) {
            const int fr = (int)first_read(c);
            sched[(fr + nsteps - 1) % nsteps] |= 1 << 16;
        }
    // the same commits, seen from tile 0, must be what tile 1 shows (periodicity): chunk g of tile 0 at step c <=> chunk g of tile 1 at c
    // (holds because first/last_read are tile-periodic and the prologue's state equals the steady state's: checked by the emulation test)
    return 1;
}

static int fpatch_geom(const mhip_conv_f32_t *p, fpatch_geom_t *g, int frames, int rec = 0) {
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
    g->rec = rec; g->nb = P_NB;
    // strip width: a divisor of out_w, multiple of 4; the one with the smallest patch (ties: the wider)
    const int woff_base = (g->nsteps * 5 * 4 + 512 + 64 + 2 * P_PRCAP * 8 + 255) & ~255; // tables | bias (record form) | zero record | row tables
    const int wbytes = 2 * 2 * g->BM * 64; // two weight stages: hi + mid planes of a K step
    int best = 0;
    for (int P_BN = g->BM <= 64 ? 512 : 256; P_BN >= 256 && !best; P_BN -= 256) // (the larger tile where its patch fits)
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
        const int cells = PR * (PWP / 4);
        const int cpi = cells * 4 <= P_NT ? 2 : (cells * 2 <= P_NT ? 4 : 8);
        const int nitems = cells * (8 / cpi);
        const int slotpix = (PR * PWP + 31) & ~31; // whole 1 KB blocks (the record form fills a slot by LDS-DMA, 1 KB per wave instruction)
        const int lds = woff_base + wbytes + P_NB * slotpix * 32;
        if (nitems > P_NT || lds > 160 * 1024) continue;
        if (rec && slotpix / 32 > 64) continue; // (8 waves x 8 DMA instructions per chunk)
        if (!best || PR * PWP < g->PR * g->PWP || (PR * PWP == g->PR * g->PWP && SW > g->SW)) {
            best = 1;
            g->SW = SW; g->dx = dx; g->PWP = PWP; g->PWH = PWP / 2; g->PR = PR; g->nitems = nitems; g->cpi = cpi; g->ngrp = PWP / 4; g->slotpix = slotpix;
            g->woff = woff_base; g->poff = woff_base + wbytes; g->lds_bytes = lds; g->bn = P_BN;
        }
    }
    if (!best) return 0;
    g->nstrips = p->out_w / g->SW;
    g->ndma = g->slotpix / 32;
    if (rec) {
        // four slots where they fit: a chunk's DMA is then issued two chunk periods earlier (one workgroup per CU by registers:
        // nothing else hides a DMA's latency)
        const int lds4 = g->lds_bytes + 2 * g->slotpix * 32;
        const bool four = g->nchunk % 4 == 0 && lds4 <= 160 * 1024;
        if (four) { g->nb = 4; g->lds_bytes = lds4; }
        else return 0; // two slots: a chunk's DMA would be issued one chunk period before its first reader -- measured SLOWER than the
                       // register-staged form on the stride-2 layers whose patch leaves no room for four (3 x 3 s2 128 -> 256 @80: 1226 vs 1067 us)
        if (g->nsteps < 4 || (size_t)2 * g->oc_pad * g->kp * 2 > 0x7ffffff0ull) return 0;
    }
    int sched[1024];
    if (g->nsteps > 1024 || !fpatch_schedule(g->nchunk, g->U, g->nsteps, sched, g->nb, rec)) return 0;
    const long total = (long)frames * p->out_h * p->out_w;
    const size_t in_bytes = (size_t)(frames - 1) * p->in_stride + (size_t)p->in_c * p->in_h * p->in_w * 4;
    if (total > 0x7fffffffL - g->bn || in_bytes > 0xfffffff0ull || (size_t)frames * p->out_stride > 0xfffffff0ull) return 0;
    g->total_pix = (unsigned)total;
    g->ntiles = (unsigned)((total + g->bn - 1) / g->bn);
    g->nsegs = (unsigned)(frames * g->nstrips);
    g->in_bytes = (unsigned)in_bytes;
    g->out_bytes = (unsigned)((size_t)(frames - 1) * p->out_stride + (size_t)p->out_c * p->out_h * p->out_w * 4);
    g->dSW = make_pdiv((unsigned)g->SW); g->dHo = make_pdiv((unsigned)g->H_out); g->dHV = make_pdiv((unsigned)g->HV);
    g->dNS = make_pdiv((unsigned)g->nstrips); g->dgrp = make_pdiv((unsigned)g->ngrp); g->dPWP = make_pdiv((unsigned)g->PWP);
    return 1;
}

// patch-pixel offset of unit (chunk c, tap) of the K stream: ring slot + tap position (stride 2: de-interleaved columns)
static int fpatch_toff(const fpatch_geom_t *g, int c, int tap) {
    const int ky = tap / g->kw, kx = tap - ky * g->kw, v = g->dx + kx;
    return (c % g->nb) * g->slotpix + ky * g->PWP + (g->s == 2 ? (v >> 1) + (v & 1) * g->PWH : v);
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

// Bytes of, and (w, out != NULL) the content of, the image conv_f32_patch reads: [nsteps][4] unit offsets (bytes into the patch ring, -1 = dummy), [nsteps] schedule, then
// two planes (hi, mid) of bf16 [oc_pad][kp] in the kernel's K order: element 8 u + j of a row = channel 8 c + j, tap of unit u =
// (chunk c, tap) (dummy units and the slack: zeros).  0 = not a shape this kernel takes.
// rec != 0: the image of the record-input form (conv_f32_prec): the ring may have four slots, the schedule names DMA issue steps.
extern "C" size_t mhip_conv_f32_patch_pack2(int out_c, int in_c, int kh, int kw, int stride, int pad, int in_h, int in_w, int out_h, int out_w,
                                            int rec, const float *w, void *out) {
    mhip_conv_f32_t p;
    shape_of(&p, out_c, in_c, kh, kw, stride, pad, in_h, in_w, out_h, out_w);
    fpatch_geom_t g;
    if (out_c <= 0 || !fpatch_geom(&p, &g, 1, rec != 0)) return 0;
    const size_t tabb = ((size_t)g.tab_ints * 4 + 255) & ~(size_t)255, planeb = (size_t)g.oc_pad * g.kp * 2;
    const size_t bytes = tabb + 2 * planeb;
    if (!w || !out) return bytes;
    memset(out, 0, bytes);
    int *tabs = (int *)out;
    int units[4096];
    fpatch_units(g.nchunk, g.U, units, 4096);
    for (int i = 0; i < g.nsteps * 4; i++) tabs[i] = units[i] < 0 ? -1 : fpatch_toff(&g, units[i], i - units[i] * g.U) * 32; // bytes: 32 per patch pixel
    fpatch_schedule(g.nchunk, g.U, g.nsteps, tabs + g.nsteps * 4, g.nb, g.rec);
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

extern "C" size_t mhip_conv_f32_patch_pack(int out_c, int in_c, int kh, int kw, int stride, int pad, int in_h, int in_w, int out_h, int out_w,
                                           const float *w, void *out) {
    return mhip_conv_f32_patch_pack2(out_c, in_c, kh, kw, stride, pad, in_h, in_w, out_h, out_w, 0, w, out);
}

// the geometry as ints (tests / tools): s kh kw pad C nchunk U SW nstrips H_in W_in H_out W_out HV PR PWP PWH dx slotpix nsteps ngrp
// nitems BM kp oc_pad tab_ints ndummy cpi bn woff poff lds_bytes nb rec ndma; returns how many were written (0 = not a shape this kernel takes)
extern "C" int mhip_conv_f32_patch_geom2(int out_c, int in_c, int kh, int kw, int stride, int pad, int in_h, int in_w, int out_h, int out_w, int rec,
                                         int *outv, int cap) {
    mhip_conv_f32_t p;
    shape_of(&p, out_c, in_c, kh, kw, stride, pad, in_h, in_w, out_h, out_w);
    fpatch_geom_t g;
    if (out_c <= 0 || !fpatch_geom(&p, &g, 1, rec != 0)) return 0;
    const int n = 35;
    if (cap < n) return 0;
    memcpy(outv, &g, n * sizeof(int));
    return n;
}
extern "C" int mhip_conv_f32_patch_geom(int out_c, int in_c, int kh, int kw, int stride, int pad, int in_h, int in_w, int out_h, int out_w, int *outv, int cap) {
    return mhip_conv_f32_patch_geom2(out_c, in_c, kh, kw, stride, pad, in_h, in_w, out_h, out_w, 0, outv, cap);
}

// which form reads this layer's input when it arrives as records: 0 none, 1 conv_f32_prec (four ring slots, LDS-DMA; the image must be
// packed with rec = 1), 2 conv_f32_patch<..., RECIN> (two slots, through registers; the plain image)
extern "C" int mhip_conv_f32_patch_rec_form(int out_c, int in_c, int kh, int kw, int stride, int pad, int in_h, int in_w, int out_h, int out_w) {
    mhip_conv_f32_t p;
    shape_of(&p, out_c, in_c, kh, kw, stride, pad, in_h, in_w, out_h, out_w);
    fpatch_geom_t g;
    if (out_c <= 0) return 0;
    if (fpatch_geom(&p, &g, 1, 1)) return 1;
    if (!fpatch_geom(&p, &g, 1, 0)) return 0;
    return g.bn == 256 && (g.slotpix * 2 + P_NT - 1) / P_NT <= 8 ? 2 : 0;
}

static unsigned long g_patch_launches = 0, g_prec_launches = 0, g_recin_launches = 0;
extern "C" unsigned long mhip_conv_f32_recin_launches(void) { return g_recin_launches; }
extern "C" unsigned long mhip_conv_f32_patch_launches(void) { return g_patch_launches; }
extern "C" unsigned long mhip_conv_f32_prec_launches(void) { return g_prec_launches; }

// The grid is (pixel runs, channel tiles), workgroups are dealt to the 8 XCDs round-robin in dispatch order (x fastest): the channel tiles of
// one pixel run share an XCD -- and its L2, where the second, third ... tile's input reads should hit -- only if the x extent is a multiple
// of 8.  (It was ceil(tiles / per): 58 for the 400 tiles of a 20 x 20 map over 64 slots -- the four channel tiles of a run sat on four XCDs
// and each fetched the input itself.  PMC, config 5 per batch: conv_f32_patch<128,...>'s four float-input launches 8.2 -> 4.6 GB read,
// conv_f32_split<128,...> 19.8 -> 11.5 GB; the TIME of those launches did not move -- the Infinity Cache had been serving the repeats.)
static unsigned fpatch_grid_x(unsigned gx, unsigned cap, unsigned noc) {
    if (noc > 1 && gx >= 8) {
        const unsigned up = (gx + 7) & ~7u;
        gx = up <= cap ? up : (gx & ~7u);
    }
    return gx;
}

template <int BM, int WM, int WN, int CPI, bool DUMMY, int BN, int RECIN = 0>
static int launch_patch(const mhip_conv_f32_t *p, fpatch_geom_t &g) {
    auto kern = conv_f32_patch<BM, WM, WN, CPI, DUMMY, BN, RECIN>;
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
    gx = fpatch_grid_x((g.ntiles + per - 1) / per, (unsigned)(slots > 0 ? slots : cus) / noc, noc);
    g.per = per;
    const size_t tabb = ((size_t)g.tab_ints * 4 + 255) & ~(size_t)255;
    hipLaunchKernelGGL(kern, dim3(gx, noc), dim3(P_NT), (size_t)g.lds_bytes, mhip_stream_native(), *p, g, (const int *)p->w_patch,
                       (const int8_t *)p->w_patch + tabb);
    if (RECIN) g_recin_launches++;
    else g_patch_launches++;
    return mhip_check(hipGetLastError(), "conv_f32_patch");
}

template <int BM, int WM, int WN, bool DUMMY, int BN, int NDW>
static int launch_prec(const mhip_conv_f32_t *p, fpatch_geom_t &g) {
    auto kern = conv_f32_prec<BM, WM, WN, DUMMY, BN, NDW>;
    static int cus = 0;
    if (!cus) {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return mhip_check(hipErrorUnknown, "conv_f32_prec attribute");
        cus = prop.multiProcessorCount;
    }
    const unsigned noc = (unsigned)((p->out_c + BM - 1) / BM);
    if (noc > 65535u) return -2;
    int slots = 0;
    mhip_conv_i8_tune_get("persist_slots", &slots);
    unsigned gx = (unsigned)(slots > 0 ? slots : cus) / noc; // one 8-wave workgroup per CU (registers)
    if (gx < 1) gx = 1;
    if (gx > g.ntiles) gx = g.ntiles;
    const unsigned per = (g.ntiles + gx - 1) / gx;
    gx = fpatch_grid_x((g.ntiles + per - 1) / per, (unsigned)(slots > 0 ? slots : cus) / noc, noc);
    g.per = per;
    const size_t tabb = ((size_t)g.tab_ints * 4 + 255) & ~(size_t)255;
    hipLaunchKernelGGL(kern, dim3(gx, noc), dim3(P_NT), (size_t)g.lds_bytes, mhip_stream_native(), *p, g, (const int *)p->w_patch,
                       (const int8_t *)p->w_patch + tabb);
    g_prec_launches++;
    return mhip_check(hipGetLastError(), "conv_f32_prec");
}

// -2: not a shape this kernel takes (the caller goes on to conv_f32_split), else the launch result.  in_rec: the input is in record
// format -- 1: conv_f32_prec, the image (w_patch) was packed for that form (mhip_conv_f32_patch_pack2 with rec = 1); 2: conv_f32_patch's
// record-input form, the plain image (mhip_conv_f32_patch_rec_form says which one a shape takes: the planner's pairing)
int conv_f32_try_patch(const mhip_conv_f32_t *p) {
    if (!p->w_patch || p->use_mfma != 3 || p->out_rec) return -2;
    fpatch_geom_t g;
    if (!fpatch_geom(p, &g, p->frames, p->in_rec == 1)) return -2; // in_rec 1: the four-slot DMA form (its image); 2: record input through registers (the plain image)
    if (p->add && p->add_stride != p->out_stride) return -2;
    if (p->in_rec == 1) {
        const int ndw = (g.ndma + 7) / 8;
#define FR_N(BM, WM, WN, D, BN) (ndw <= 2 ? launch_prec<BM, WM, WN, D, BN, 2>(p, g) : ndw <= 3 ? launch_prec<BM, WM, WN, D, BN, 3>(p, g) : ndw <= 4 ? launch_prec<BM, WM, WN, D, BN, 4>(p, g) : ndw <= 6 ? launch_prec<BM, WM, WN, D, BN, 6>(p, g) : launch_prec<BM, WM, WN, D, BN, 8>(p, g))
#define FR_D(BM, WM, WN, BN) (g.ndummy ? FR_N(BM, WM, WN, true, BN) : FR_N(BM, WM, WN, false, BN))
        if (g.BM == 128) return FR_D(128, 2, 4, 256);
        if (g.BM == 64) return g.bn == 512 ? FR_D(64, 1, 8, 512) : FR_D(64, 1, 8, 256);
        return g.bn == 512 ? FR_D(32, 1, 8, 512) : FR_D(32, 1, 8, 256);
#undef FR_D
#undef FR_N
    }
    if (p->in_rec) { // four ring slots do not fit: record input through registers, two slots (conv_f32_patch<..., RECIN>); 256-pixel tiles only
        const int nri = (g.slotpix * 2 + P_NT - 1) / P_NT;
        if (g.bn != 256 || nri > 8) return -2;
#define FQ_N(BM, WM, WN, D) (nri <= 4 ? launch_patch<BM, WM, WN, 8, D, 256, 4>(p, g) : nri <= 6 ? launch_patch<BM, WM, WN, 8, D, 256, 6>(p, g) : launch_patch<BM, WM, WN, 8, D, 256, 8>(p, g))
#define FQ_D(BM, WM, WN) (g.ndummy ? FQ_N(BM, WM, WN, true) : FQ_N(BM, WM, WN, false))
        if (g.BM == 128) return FQ_D(128, 2, 4);
        if (g.BM == 64) return FQ_D(64, 1, 8);
        return FQ_D(32, 1, 8);
#undef FQ_D
#undef FQ_N
    }
#define FP_D(BM, WM, WN, CPI, BN) (g.ndummy ? launch_patch<BM, WM, WN, CPI, true, BN>(p, g) : launch_patch<BM, WM, WN, CPI, false, BN>(p, g))
#define FP_CPI(BM, WM, WN, BN) (g.cpi == 8 ? FP_D(BM, WM, WN, 8, BN) : g.cpi == 4 ? FP_D(BM, WM, WN, 4, BN) : FP_D(BM, WM, WN, 2, BN))
    if (g.BM == 128) return FP_CPI(128, 2, 4, 256);
    if (g.BM == 64) return g.bn == 512 ? FP_CPI(64, 1, 8, 512) : FP_CPI(64, 1, 8, 256);
    return g.bn == 512 ? FP_CPI(32, 1, 8, 512) : FP_CPI(32, 1, 8, 256);
#undef FP_CPI
#undef FP_D
}
