// conv_i8_rows.hip -- the deep 3x3 stride-1 convolutions (in_c a multiple of 128, 128-channel output tiles, 20- / 40-wide maps)
// as ONE persistent 8-wave workgroup per CU that walks tiles of 256 consecutive map pixels (reference
// src/mars/mxu_conv.c:713-757; same arithmetic contract as conv_i8.hip).  Round 4, verdict item 1.  Launch variant 20.
//
// Why another form.  The implicit-GEMM tile (conv_i8_mfma<256,128,3,1,8>) moves (256 + 128) x 64 bytes into LDS per K step
// for 512 CU-cycles of matrix work = 48 B/clk/CU, and the L2 -> LDS path delivers 20-22 under every loop structure tried
// (DESIGN.md section 5 "Deep-K"): 0.45 busy in the loop, 0.34 with the ~35 % of a launch that is per-tile fixed cost.  Here
//   * the input is staged as a PATCH, 64 channels at a time (two buffers: the chunk being multiplied and the next one): each
//     byte enters LDS once per tile instead of once per tap -- the nine taps of a K chunk are nine LDS addresses;
//   * a tile is 256 CONSECUTIVE pixels of the map in row-major order, frames stacked (6.4 rows of a 40-wide map, 12.8 of a
//     20-wide one), so every MFMA column carries 16 real pixels whatever the map's size: round 1's 16 x 16 tiles covered
//     69 % / 39 % of their pixels on the 40 x 40 / 20 x 20 maps of a 640 x 640 input.  The patch holds the rows the tile
//     touches plus the halo; a tile may straddle two frames, the patch then holds the zero rows between them (rowtab: one
//     entry per patch row = byte offset of that input row, or -1 = zeros);
//   * the weights of a K step (128 x 64 bytes, pre-laid on the host in LDS image order: mhip_conv_i8_rows_pack) stream
//     through a 9-slot ring (slot = tap), RW_AHEAD = 6 steps ahead of their readers, as whole 128-byte lines: one 1 KB
//     `buffer_load ... lds` per wave and step; 8 KB + ~3 KB of patch per 512 CU-cycles = 22 B/clk/CU;
//   * one workgroup per CU owns a run of tiles: index math, tables and the ring's prologue are paid once per run.
// The shipped loop (what the code below does; the alternatives that were built and measured slower are in
// profiles/r04_experiments.md section 1): the K stream runs in PAIRS of steps, two 64-channel chunks (18 steps) per loop
// iteration; a pair is a READING phase (the pair's 16 fragment reads + its DMA issue) and a MULTIPLYING phase (32 MFMAs), one
// raw s_barrier after each; waves 4-7 (the second wave of each SIMD) run ONE PHASE behind waves 0-3, so on every SIMD one
// wave multiplies while its mate reads.  The epilogue of a tile runs IN LINE after the tile's last pair (all 64 values of a
// lane requantised / looked up in place in the accumulator registers, four 16-byte buffer stores per lane).
// vmcnt invariants (loads, LDS-DMA and stores retire in order against one counter on gfx9):
//   * prologue: [patch pieces] W0..W5 issued; vmcnt(4) => patch, W0, W1 of this wave landed; barrier => everybody's.
//   * reading phase of pair j issues [<= 2 patch pieces] W(2j+6) W(2j+7) and then waits vmcnt(4): at most 4 operations stay
//     in flight -- the two weight blocks just issued and the two of pair j-1 (patch pieces, issued before the weights of
//     their pair, are older than those and have landed).  So W(2j+2), W(2j+3) -- what pair j+1 reads -- are in LDS one
//     barrier before any wave reads them.
//   * behind a tile's epilogue the lane's 4 stores sit between the weight blocks in that in-order count.  Pair 0 of the next
//     tile waits vmcnt(10) (in flight: W4 W5 | S S S S | [P P] W6 W7 -- the stores and everything younger), pair 1 waits
//     vmcnt(8) (S S S S | [P P] W6 W7 | [P P] W8 W9 minus the two patch pieces that may not exist: 8 is exact when both
//     pairs carry patch pieces and conservative -- waits for more -- when they do not), so the stores are never waited for
//     in the K stream (waiting cost ~7 us per tile: every workgroup of the launch stores at the same moment).
//   * a ring slot is rewritten one and a half pairs after its last reader's fragments were in registers (lgkmcnt(0) before
//     the reading phase's barrier).
#include "conv_i8_common.hpp"

#define RW_NPX 256                 // pixels per tile
#define RW_COLS 4                  // MFMA pixel columns per wave (4 pixel groups x 4 x 16 = 256)
#define RW_BN 128                  // output channels per workgroup
#define RW_RING 9                  // weight ring stages = the nine taps of a chunk (slot = tap), RW_AHEAD of them in flight
#define RW_AHEAD 6
#define RW_STAGE (RW_BN * BK)      // 8 KB
#define RW_NDWMAX 6                // patch DMA instructions per wave and chunk, at most
#define RW_FIXED (2048 + RW_RING * RW_STAGE) // LUT | bias | rowtab | ring

#ifndef ROWS_ABL
#define ROWS_ABL 0 // ablation builds (timing only, wrong bytes): bit 0 no patch DMA in the K stream, bit 1 no weight DMA,
                   // bit 2 no epilogue, bit 3 no MFMA, bit 4 / 5 trailing waves = odd waves / waves 2,3,6,7, bit 6 no vmcnt wait in the K stream, bit 7 no epilogue but live accumulators
#endif
#ifdef ROWS_STAMPS // diagnostic build only (tools/stamps_build.sh): where a K step's cycles go, per wave class
__device__ unsigned long long rows_stamp_sums[16];
extern "C" int mhip_rows_stamps(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(rows_stamp_sums), sizeof(rows_stamp_sums)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(rows_stamp_sums), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#define RSTAMP(t)                                                                         \
    do {                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                \
    } while (0)
#else
#define RSTAMP(t) do { } while (0)
#endif

// patch DMA instructions per wave and chunk the LDS budget allows at this patch pitch (x 8 KB = one of the two patch buffers)
constexpr int rows_patch_pieces(int pwp) { return pwp <= 24 ? 4 : (pwp <= 48 ? 5 : 6); }

struct rows_args_t {
    int W, H;          // map width / height
    int C, nchunk;     // input channels, 64-channel chunks (>= 2)
    int prmax;         // patch rows (<= 64)
    int ndw;           // patch DMA instructions per wave and chunk
    unsigned ntiles, noc, ngrp, nblk;
    unsigned npix;     // frames * H * W
    unsigned in_bytes, out_bytes, w_bytes;
    fastdiv_t dW, dH, dH2;
};

// PWP = patch row pitch in pixels = roundup8(W + 2): a multiple of 8 keeps the bank swizzle (unit bit 1 ^= pixel bit 2)
// invariant under the kernel-row offsets, which therefore sit in the ds_read's immediate; kernel columns 0..2 have
// their own precomputed address each.
template <int PWP, bool HAS_LUT>
__global__ __launch_bounds__(512) void conv_i8_rows(const mhip_conv_i8_t p, const rows_args_t a) {
    ANAT_BEGIN();
    extern __shared__ __attribute__((aligned(16))) int8_t dynlds[];
    uint8_t *slut = (uint8_t *)dynlds; // LDS byte address 0 (LUT look-ups use immediate offsets)
    lds_base_must_be_zero(dynlds);
    int *sbias = (int *)(dynlds + 512);
    int *rowtab = (int *)(dynlds + 1024); // [2][64]
    int8_t *ring = dynlds + 2048;
    constexpr int PATCH0 = RW_FIXED;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv & 3, wn = wv >> 2;
#if ROWS_ABL & 16
    const bool late = (wv & 1) != 0;
#elif ROWS_ABL & 32
    const bool late = (wv & 2) != 0;
#else
    const bool late = wv >= 4; // the second wave of its SIMD (waves are dealt to SIMDs 0,2,1,3,0,2,1,3): runs one phase behind
#endif
    const int frow = lane & 15, fchunk = lane >> 4;
    const unsigned id = xcd_remap(blockIdx.x, a.nblk);
    const unsigned grp = id / a.noc, ot = id - grp * a.noc;
    const int oc0 = (int)ot * RW_BN;
    const unsigned t0 = (unsigned)(((unsigned long long)grp * a.ntiles) / a.ngrp);
    const unsigned t1 = (unsigned)(((unsigned long long)(grp + 1) * a.ntiles) / a.ngrp);
    if (t0 >= t1) return;
    const int W = a.W, H = a.H, C = a.C;
    constexpr int PB_C = rows_patch_pieces(PWP) * 8 * 1024; // bytes of one patch buffer
    const int NK = a.nchunk * 9;

    if (HAS_LUT && tid < 128) ((uint32_t *)slut)[tid] = ((const uint32_t *)p.lut2)[tid];
    if (tid < RW_BN) sbias[tid] = p.bias ? p.bias[oc0 + tid] : 0;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w_rows, 0, (int)a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)a.out_bytes, 0x00020000);
    const int pstride = p.out_pix_stride ? p.out_pix_stride : p.out_c;
    const int chan = wn * 64 + fchunk * 16; // first of this lane's 16 consecutive channels (tile-relative)
    const float cs2 = p.cs * 2.0f;
    const int lo = p.relu ? 0 : -128;
    const int wvoff = wv * 1024 + lane * 16;
    const unsigned wblk0 = ot * (unsigned)NK;

    // ---- per-tile tables
    auto fill_rowtab = [&](unsigned t) __attribute__((always_inline)) { // byte offset of every patch row of tile t (wave 0, lanes < prmax)
        if (tid < a.prmax) {
            const unsigned g0 = fdiv(t * (unsigned)RW_NPX, a.dW), f0 = fdiv(g0, a.dH), y0 = g0 - f0 * (unsigned)H;
            const unsigned pr = f0 * (unsigned)(H + 2) + y0 + (unsigned)tid;
            const unsigned f = fdiv(pr, a.dH2), yy = pr - f * (unsigned)(H + 2);
            const int iy = (int)yy - 1;
            const bool ok = iy >= 0 && iy < H && f < (unsigned)p.frames;
            rowtab[(t & 1) * 64 + tid] = ok ? (int)(f * (unsigned)p.in_stride) + iy * W * C : -1;
        }
    };
    int voffp[RW_NDWMAX]; // per-lane source offsets of the patch DMA instructions (tile being loaded)
    auto patch_setup = [&](unsigned t) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < RW_NDWMAX; j++) {
            voffp[j] = -1;
            if (j < a.ndw) {
                const unsigned phys = (unsigned)((j * 8 + wv) * 64 + lane);
                const unsigned U = phys ^ ((phys >> 3) & 2u);
                const unsigned pix = U >> 2, c16 = U & 3u;
                const unsigned prow = pix / (unsigned)PWP, pcol = pix - prow * (unsigned)PWP;
                const int ro = prow < (unsigned)a.prmax ? rowtab[(t & 1) * 64 + prow] : -1;
                const bool ok = ro >= 0 && pcol >= 1u && pcol <= (unsigned)W;
                voffp[j] = ok ? ro + ((int)pcol - 1) * C + (int)c16 * 16 : -1;
            }
        }
    };
    int addr[RW_COLS][3]; // LDS byte address of this lane's B fragment in patch buffer 0: column u, kernel column kx, kernel row 0
    auto cur_setup = [&](unsigned t) __attribute__((always_inline)) {
        const unsigned P0 = t * (unsigned)RW_NPX;
        const unsigned g0 = fdiv(P0, a.dW), f0 = fdiv(g0, a.dH), y0 = g0 - f0 * (unsigned)H;
#pragma unroll
        for (int u = 0; u < RW_COLS; u++) {
            const unsigned P = P0 + (unsigned)(wm * 64 + u * 16 + frow); // this lane's pixel of column u (B operand and result)
            const unsigned g = fdiv(P, a.dW), x = P - g * (unsigned)W;
            const unsigned tr = g - g0, cross = fdiv(y0 + tr, a.dH);
            const unsigned pix = (tr + 2u * cross) * (unsigned)PWP + x;
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                const unsigned px = pix + (unsigned)kx;
                addr[u][kx] = PATCH0 + (int)(px * 64u) + ((fchunk * 16) ^ (int)((px & 4u) << 3));
            }
        }
    };

    // ---- epilogue of a tile: 64 values per lane, eight slices of 8 (column u, channel half h), in line after the tile's last
    // K step.  (Round 4 also built it INSIDE the next tile's K loop -- a second accumulator set drained slice by slice in the
    // MFMAs' issue shadow: 250+ registers, spills as soon as anything else was pinned, and only 4 % faster, because what
    // bounds this kernel is the fixed cost per barrier interval, not the epilogue: DESIGN.md section 5.)
    typedef v4i acc_t[4][RW_COLS];
    acc_t cur;
    int voff[RW_COLS];
    auto epilogue_tile = [&](unsigned t) __attribute__((always_inline)) {
        const unsigned P0 = t * (unsigned)RW_NPX;
#pragma unroll
        for (int u = 0; u < RW_COLS; u++) {
            const unsigned P = P0 + (unsigned)(wm * 64 + u * 16 + frow);
            const unsigned g = fdiv(P, a.dW), x = P - g * (unsigned)W;
            const unsigned f = fdiv(g, a.dH), y = g - f * (unsigned)H;
            const bool ok = P < a.npix && oc0 + chan < p.out_c;
            voff[u] = ok ? (int)(f * (unsigned)p.out_stride + (y * (unsigned)W + x) * (unsigned)pstride + (unsigned)(p.out_ch_off + oc0 + chan)) : -1;
        }
        // all 64 values of the lane at once, in place in the accumulator registers: index -> (one wait) -> table byte -> pack
        if (HAS_LUT) {
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int u = 0; u < RW_COLS; u++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int k = (int)((float)cur[q][u][r] * cs2);
                        cur[q][u][r] = k < -256 ? -256 : (k > 255 ? 255 : k); // the half-step table's index
                    }
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int u = 0; u < RW_COLS; u++)
                    asm volatile("ds_read_i8 %0, %0 offset:256\n\tds_read_i8 %1, %1 offset:256\n\tds_read_i8 %2, %2 offset:256\n\tds_read_i8 %3, %3 offset:256"
                                 : "+v"(cur[q][u][0]), "+v"(cur[q][u][1]), "+v"(cur[q][u][2]), "+v"(cur[q][u][3])
                                 :
                                 : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int u = 0; u < RW_COLS; u++) asm volatile("" : "+v"(cur[q][u][0]), "+v"(cur[q][u][1]), "+v"(cur[q][u][2]), "+v"(cur[q][u][3])); // after the wait
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int u = 0; u < RW_COLS; u++)
#pragma unroll
                    for (int r = 0; r < 4; r++) cur[q][u][r] = requant_safe(cur[q][u][r], p.cs, lo, 127);
        }
#pragma unroll
        for (int u = 0; u < RW_COLS; u++) {
            uint32_t pk[4];
#pragma unroll
            for (int q = 0; q < 4; q++) pk[q] = pack4(cur[q][u][0], cur[q][u][1], cur[q][u][2], cur[q][u][3]);
            __builtin_amdgcn_raw_buffer_store_b128((v4i){(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]}, ors, voff[u], 0, 0);
        }
    };

    // ---- prologue: tables and patch of the first tile, the first weight tiles
    fill_rowtab(t0);
    __syncthreads();
    patch_setup(t0);
#pragma unroll
    for (int j = 0; j < RW_NDWMAX; j++)
        if (j < a.ndw) blds16(xrs, voffp[j], 0, dynlds + PATCH0 + (j * 8 + wv) * 1024);
#pragma unroll
    for (int i = 0; i < RW_AHEAD; i++) // NK >= 18 > RW_AHEAD
        blds16(wrs, wvoff, (int)((wblk0 + (unsigned)i) * RW_STAGE), ring + i * RW_STAGE + wv * 1024);
    wait_vmcnt<RW_AHEAD - 2>();   // patch, W(0), W(1) of this wave have landed
    __builtin_amdgcn_s_barrier(); // ... and everybody else's
    asm volatile("" ::: "memory");
    // The K stream runs in PAIRS of steps, two 64-channel chunks (18 steps, 9 pairs) per loop iteration -- the first chunk in
    // patch buffer 0, the second in buffer 1, so every LDS address is an immediate.  A pair is a reading phase (16 fragment
    // reads, the pair's DMA) and a multiplying phase (32 MFMAs), a barrier after each.  The two waves of a SIMD
    // run ONE PHASE apart (waves 4-7 take one extra barrier here, waves 0-3 one at the end): between any two barriers one
    // wave of every SIMD multiplies while its partner reads.  Two steps per phase because the fixed cost of a barrier interval
    // (~250 cycles: the barrier, scalar bookkeeping) was what bounded the one-step form -- 30 us of a 86 us launch with every
    // MFMA, DMA and epilogue instruction compiled out (tools/stamps_build.sh abl N).
    // DMA is issued in the READING phase (a DMA instruction costs its wave ~100 cycles of issue: free while the partner
    // multiplies, 18 us per launch when it sat between the MFMAs).  vmcnt: reading phase j issues [patch pieces] W(2j + 6)
    // W(2j + 7) (at the run's end: into slots nobody reads again, so the count stays static) and then waits for "all but the
    // 4 youngest": left in flight are at most the weight DMAs of pairs j - 1 and j; pair j + 1's weights (issued in pair
    // j - 2), every patch piece before them and the last epilogue's stores have landed -- two barriers before the first wave
    // reads them.  A ring slot is rewritten one and a half pairs after its last reader's fragments were in registers.
    if (late) __builtin_amdgcn_s_barrier();
#ifdef ROWS_STAMPS
    unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, sum_wait = 0, sum_pre = 0, sum_mfma = 0, sum_steps = 0;
#endif

    ANAT_NOW(1);
    for (unsigned t = t0; t < t1; t++) {
        cur_setup(t);
        const bool last_tile = t + 1 == t1;
        for (int c2 = 0; c2 < a.nchunk; c2 += 2) {
            const bool last_c2 = c2 + 2 == a.nchunk;
#pragma unroll
            for (int pr = 0; pr < 9; pr++) {
                RSTAMP(s0);
                // ---- reading phase: the fragments of steps 2 pr and 2 pr + 1
                v4i xb[2][RW_COLS], wa[2][4];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int st = 2 * pr + e, tap = st % 9, half = st / 9;
                    const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                    for (int u = 0; u < RW_COLS; u++) xb[e][u] = *(const v4i *)(dynlds + addr[u][kx] + ky * (PWP * 64) + half * PB_C);
                    const int8_t *ws = ring + tap * RW_STAGE;
#pragma unroll
                    for (int q = 0; q < 4; q++) wa[e][q] = *(const v4i *)(ws + lds_off(wn * 64 + q * 16 + frow, fchunk));
                }
                // patch pieces: pairs 0-2 fill buffer 1 with this iteration's second chunk, pairs 5-7 buffer 0 with the chunk after
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int j = pr < 5 ? 2 * pr + e : 2 * (pr - 5) + e;
                    if ((pr < 3 || (pr >= 5 && pr < 8)) && j < RW_NDWMAX && !(ROWS_ABL & 1)) {
                        if (pr < 3) {
                            if (j < a.ndw) blds16(xrs, voffp[j], (c2 + 1) * 64, dynlds + PATCH0 + PB_C + (j * 8 + wv) * 1024);
                        } else {
                            if (j < a.ndw && !(last_c2 && last_tile)) blds16(xrs, voffp[j], last_c2 ? 0 : (c2 + 2) * 64, dynlds + PATCH0 + (j * 8 + wv) * 1024);
                        }
                    }
                }
                // the weight tiles of steps + 6 and + 7 (slots last read a pair and a half ago)
                if (!(ROWS_ABL & 2)) {
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        int gs = c2 * 9 + 2 * pr + RW_AHEAD + e;
                        gs = gs < NK ? gs : gs - NK;
                        blds16(wrs, wvoff, (int)((wblk0 + (unsigned)gs) * RW_STAGE), ring + ((2 * pr + RW_AHEAD + e) % RW_RING) * RW_STAGE + wv * 1024);
                    }
                }
                RSTAMP(s1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // fragments in registers before the multiplying phase starts
                // next pair's operands (this wave's share) have landed.  Behind a tile's epilogue its four stores sit between the
                // weight DMAs in this in-order count: waiting them out cost ~7 us per tile (every workgroup of the launch stores at
                // the same moment); the first two pairs of a tile therefore leave them in flight as well
                if (pr == 0 && c2 == 0 && t > t0) wait_vmcnt<10>();      // W4 W5 | S S S S | [P P] W6 W7
                else if (pr == 1 && c2 == 0 && t > t0) wait_vmcnt<8>();  // S S S S | [P P] W6 W7 | [P P] W8 W9: 8 youngest stay (header)
                else if (!(ROWS_ABL & 64)) wait_vmcnt<4>();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                RSTAMP(s2);
                // ---- multiplying phase
                // the next tile's tables while this tile's last two chunks run (barriers between writer and readers)
                if (pr == 3 && last_c2 && !last_tile) fill_rowtab(t + 1);
                if (pr == 4 && last_c2 && !last_tile) patch_setup(t + 1);
                const bool first_step = c2 == 0 && pr == 0;
                if (pr == 0 && first_step) { // a tile's first step: the bias is the C operand
                    v4i bias4[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) bias4[q] = *(const v4i *)(sbias + wn * 64 + q * 16 + fchunk * 4);
#pragma unroll
                    for (int q = 0; q < 4; q++)
#pragma unroll
                        for (int u = 0; u < RW_COLS; u++) cur[q][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[0][q], xb[0][u], bias4[q], 0, 0, 0);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; q++)
#pragma unroll
                        for (int u = 0; u < RW_COLS; u++)
                            if (!(ROWS_ABL & 8)) cur[q][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[0][q], xb[0][u], cur[q][u], 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int u = 0; u < RW_COLS; u++)
                        if (!(ROWS_ABL & 8)) cur[q][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[1][q], xb[1][u], cur[q][u], 0, 0, 0);
                RSTAMP(s3);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                RSTAMP(s4);
#ifdef ROWS_STAMPS
                sum_pre += s1 - s0; sum_wait += (s2 - s1) + (s4 - s3); sum_mfma += s3 - s2; sum_steps += 1;
#endif
            }
        }
        if (ROWS_ABL & 128) { // ablation: no epilogue, but the accumulators stay alive (one folded value, stored if it hits a magic number)
            int fold = 0;
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int u = 0; u < RW_COLS; u++) fold ^= cur[q][u][0] ^ cur[q][u][1] ^ cur[q][u][2] ^ cur[q][u][3];
            if (fold == 0x12345678) p.out[0] = 1;
        } else if (!(ROWS_ABL & 4)) epilogue_tile(t);
    }
    if (!late) __builtin_amdgcn_s_barrier(); // pairs with the trailing waves' last one
    ANAT_NOW(2);
    ANAT_END(p);
#ifdef ROWS_STAMPS
    if (lane == 0) {
        const int o = late ? 8 : 0;
        atomicAdd(&rows_stamp_sums[o + 0], sum_wait); atomicAdd(&rows_stamp_sums[o + 1], sum_pre); atomicAdd(&rows_stamp_sums[o + 2], sum_mfma);
        atomicAdd(&rows_stamp_sums[o + 4], sum_steps);
    }
#endif
}
ANAT_SETTER(mhip_anatomy_set_rows)

// ---------------------------------------------------------------------------------
// host side
struct rows_geom_t {
    rows_args_t a;
    int pwp;
    size_t lds;
};
static bool rows_geom(const mhip_conv_i8_t *p, rows_geom_t *g) {
    const int C = p->in_c, W = p->out_w, H = p->out_h;
    if (!conv_i8_direct_rows(p) || !p->safe || p->kh != 3 || p->kw != 3 || p->stride_h != 1 || p->stride_w != 1 || p->pad_top != 1 ||
        p->pad_left != 1 || p->in_h != H || p->in_w != W || C < 128 || (C & 63) || p->oc_pad % RW_BN != 0 || p->row_pad != 3 * C ||
        p->nseg > 1 || p->add || p->pre_w || (p->lut && !p->lut2))
        return false;
    if (W != 20 && W != 40) return false; // instantiated patch pitches (80-wide maps: two 48 KB patch buffers + the ring exceed 160 KB)
    if (in_extent_bytes(p) > 0x7fffffffL || persist_out_bytes(p) > 0x7fffffffL || (long)p->oc_pad * 9 * C > 0x7fffffffL) return false;
    rows_args_t &a = g->a;
    a.W = W; a.H = H; a.C = C; a.nchunk = C / 64;
    const int maxrows = (RW_NPX + W - 2) / W + 1;  // rows that RW_NPX consecutive pixels can touch
    a.prmax = maxrows + 2 + 2 * ((maxrows + H - 2) / H); // + halo + the two zero rows of every frame boundary inside
    g->pwp = (W + 2 + 7) & ~7;
    if (a.prmax > 64) return false;
    const long units = (long)a.prmax * g->pwp * 4;
    a.ndw = (int)((units + 511) / 512);
    if (a.ndw > rows_patch_pieces(g->pwp)) return false; // (short maps: more frame boundaries, i.e. zero rows, inside a tile)
    if (a.nchunk & 1) return false;                      // the K stream runs two chunks per loop iteration
    g->lds = RW_FIXED + 2 * (size_t)rows_patch_pieces(g->pwp) * 8 * 1024;
    if (g->lds > 160 * 1024) return false;
    const long npix = (long)p->frames * H * W;
    if (npix > 0x3fffffffL) return false;
    a.npix = (unsigned)npix;
    a.ntiles = (unsigned)((npix + RW_NPX - 1) / RW_NPX);
    a.noc = (unsigned)(p->oc_pad / RW_BN);
    a.in_bytes = (unsigned)in_extent_bytes(p);
    a.out_bytes = (unsigned)persist_out_bytes(p);
    a.w_bytes = (unsigned)((long)p->oc_pad * 9 * C);
    a.dW = make_fastdiv((unsigned)W);
    a.dH = make_fastdiv((unsigned)H);
    a.dH2 = make_fastdiv((unsigned)(H + 2));
    return true;
}

bool conv_i8_rows_ok(const mhip_conv_i8_t *p) {
    rows_geom_t g;
    return p->w_rows != nullptr && rows_geom(p, &g);
}

template <int PWP, bool HAS_LUT>
static int launch_rows_t(const mhip_conv_i8_t *p, rows_geom_t &g) {
    auto kern = conv_i8_rows<PWP, HAS_LUT>;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return mhip_check(hipErrorUnknown, "conv_i8_rows attribute");
        cus = prop.multiProcessorCount;
    }
    rows_args_t &a = g.a;
    unsigned slots = (unsigned)(conv_i8_tune_state().persist_slots > 0 ? conv_i8_tune_state().persist_slots : cus);
    unsigned ngrp = slots / a.noc;
    if (ngrp < 1) ngrp = 1;
    if (ngrp > a.ntiles) ngrp = a.ntiles;
    a.ngrp = ngrp;
    a.nblk = ngrp * a.noc;
    hipLaunchKernelGGL(kern, dim3(a.nblk), dim3(512), g.lds, mhip_stream_native(), *p, a);
    return mhip_check(hipGetLastError(), "conv_i8_rows launch");
}

static unsigned long g_rows_launches = 0; // launches of conv_i8_rows since load (tests: did a forced variant 20 take this kernel?)
extern "C" unsigned long mhip_conv_i8_rows_launches(void) { return g_rows_launches; }

int conv_i8_launch_rows(const mhip_conv_i8_t *p) {
    rows_geom_t g;
    if (!p->w_rows || !rows_geom(p, &g)) return -1;
    g_rows_launches++;
#define ROWS(P) (p->lut ? launch_rows_t<P, true>(p, g) : launch_rows_t<P, false>(p, g))
    if (g.pwp == 24) return ROWS(24);
    if (g.pwp == 48) return ROWS(48);
#undef ROWS
    return -1;
}

// Bytes of, and (out != NULL) the content of, the weight image conv_i8_rows streams: per (channel tile, 64-channel
// chunk, tap) one 8 KB block = the LDS image of that K step's 128 x 64 weight bytes (rows in the packed order, the
// 16-byte slots of a row swizzled as lds_off() reads them).  0 = not a shape that kernel takes: the same tests as
// rows_geom() -- an even number of 64-channel chunks, and (out_w != 0) one of the instantiated map widths -- so that a
// second weight image is only reserved where the kernel can use it.  out_w = 0: the map width is not known (any).
extern "C" size_t mhip_conv_i8_rows_pack(int in_c, int kh, int kw, int stride_h, int stride_w, int oc_pad, int k64, int out_w,
                                         const int8_t *packed, int8_t *out) {
    if (kh != 3 || kw != 3 || stride_h != 1 || stride_w != 1 || in_c < 128 || (in_c & 127) || oc_pad % RW_BN != 0 || k64 != 9 * in_c)
        return 0;
    if (out_w != 0 && out_w != 20 && out_w != 40) return 0;
    const size_t bytes = (size_t)oc_pad * 9 * in_c;
    if (!out || !packed) return bytes;
    const int nchunk = in_c / 64, noc = oc_pad / RW_BN;
    for (int ot = 0; ot < noc; ot++)
        for (int ch = 0; ch < nchunk; ch++)
            for (int tap = 0; tap < 9; tap++) {
                int8_t *blk = out + ((size_t)(ot * nchunk + ch) * 9 + tap) * RW_STAGE;
                for (int r = 0; r < RW_BN; r++) {
                    const int8_t *src = packed + (size_t)(ot * RW_BN + r) * k64 + (size_t)tap * in_c + ch * 64;
                    for (int slot = 0; slot < 4; slot++) {
                        const int chunk = slot ^ ((r >> 1) & 2);
                        memcpy(blk + r * 64 + slot * 16, src + chunk * 16, 16);
                    }
                }
            }
    return bytes;
}
