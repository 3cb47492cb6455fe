// conv_i8_patch.hip -- patch-staged int8 convolution for gfx950 (k x k kernels on wide feature maps).
// Replaces reference src/mars/mxu_conv.c:713-757 (conv2d_int8_nhwc_mxu) for those shapes; arithmetic contract
// and epilogue: conv_i8_common.hpp.
#include "conv_i8_common.hpp"
#include <stdio.h>

// ---------------------------------------------------------------------------------
// patch-staged kernel: k x k convolutions on wide feature maps with few channels (the 160x160 / 80x80 layers of
// yolov5: in_c 32..128).  The implicit-GEMM kernels fetch every input pixel once per kernel tap through the
// 64 B/clk L1 path, which is what bounds these layers (few output channels per fetched byte).  Here a workgroup
// stages the input patch of a TH x 16 output tile ONCE in LDS (LDS-DMA), keeps the weights of its channel tile
// resident in LDS for its whole (persistent) life, and feeds the MFMAs of all taps from LDS: HBM/L2 bytes are read
// once, the K loop has no barrier and no global access at all.
//   B operand of pixel (oy,ox), K chunk (ky,kx,c16) = patch[(oy*s+ky)][(ox*s+kx)][c16]: in NHWC a chunk never
//   straddles pixels, so its LDS address is Ubase(oy,ox) + dU(ky,kx,c16) in 16-byte units, dU tabulated per K step.
//   Stride 2: patch columns are stored de-interleaved (even columns, then odd), so 16 consecutive output pixels
//   read 16 consecutive patch pixels for every tap.  Bank conflicts: 16-byte unit U goes to U ^ ((U>>3) & M),
//   M = 0 / 2 / 6 for in_c = 32 / 64 / 128 -- with it the lane groups of ds_read_b128 touch 16 distinct
//   16-byte bank groups for any tap (derivation: DESIGN.md section 5).
//
// Round 3: the patches form a RING of 1..4 buffers.  These layers are HBM-bound, and what an HBM-bound kernel needs is
// bytes in flight (latency x bandwidth ~ 64-100 KB per CU); the round-2 form had at most one patch in flight per
// workgroup, and the compiler had quietly serialised even that: __syncthreads() drains the vector-memory counter when an
// LDS-DMA is pending, and the waits it placed for the bias / residual registers (ordinary loads) were vmcnt(0) -- the next
// tile's patch, just issued, was waited for before this tile's first MFMA, and every tile waited for the previous tile's
// stores.  Now: the patch of tile i + ring - 1 is issued when tile i starts; waits are counted by hand (every wave knows how
// many vector-memory operations it has issued since the patch it needs); barriers are raw s_barrier; no load the
// compiler counts is left inside the tile loop: the bias registers are complete before the first DMA is issued, and the
// residual operand of a fused Add travels by LDS-DMA too -- every lane's 16 (or 2 x 4) bytes into a staging row of its own
// wave, read back by the same lane after the wave's own counted wait (wave-private: no barrier).  (Ordinary loads hidden
// from the compiler in inline asm were tried first: it may copy an asm statement's destination registers at once, before
// the data has landed -- it did, in the 4-row instantiation.)
#ifdef PATCH_STAMPS // diagnostic build only (tools/stamps_build.sh patch, tools/patch_stamps.py): where a tile's cycles go, summed over all waves
__device__ unsigned long long patch_stamp_sums[8];
extern "C" int mhip_patch_stamps(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(patch_stamp_sums), sizeof(patch_stamp_sums)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(patch_stamp_sums), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#define STAMP(t)                                                                          \
    do {                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                \
    } while (0)
#else
#define STAMP(t) do { } while (0)
#endif
#define PT_TW 16
#define PT_NIMAX 10 // LDS-DMA instructions per wave and patch: patches of up to 40 KB

// s_waitcnt vmcnt(n) for a wave-uniform run-time n in [0, HI] (any smaller immediate would be correct too, only slower):
// a binary tree of scalar compares down to the immediate
template <int LO, int HI>
__device__ __forceinline__ void wait_vmcnt_tree(int n) {
    if constexpr (LO == HI) {
        wait_vmcnt<LO>();
    } else {
        constexpr int MID = (LO + HI + 1) / 2;
        if (n >= MID) wait_vmcnt_tree<MID, HI>(n);
        else wait_vmcnt_tree<LO, MID - 1>(n);
    }
}
template <int HI>
__device__ __forceinline__ void wait_vmcnt_upto(int n) {
    wait_vmcnt_tree<0, HI>(n > HI ? HI : n);
}
// one raw barrier: LDS operations of this wave complete first (a pending LDS-DMA is NOT waited for: that is the point)
__device__ __forceinline__ void barrier_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// PRE (fused C3 bottleneck, stride 1, in_c 32 / 64): t = SiLU(conv1x1(x)) is evaluated on the staged patch of x -- halo
// included, zero where the pixel lies outside the image (the k x k convolution's SAME padding applies to t) -- into a
// second patch buffer, and the K loop reads that one: t never goes to HBM.  The 1x1's weights ([in_c][64], K padded
// with zeros) and its half-step table (LDS bytes 512..1023) stay resident like the main weights.
template <int TH, int BN, bool HAS_LUT, bool PRE = false>
__global__ __launch_bounds__(NTHREADS) void conv_i8_patch(const mhip_conv_i8_t p, const int k64, const int tiles_x,
                                                          const int tiles_y, const unsigned ntiles_all, const int PH,
                                                          const int PW, const int PWP, const int PWH, const int nblk,
                                                          const int8_t *__restrict__ zeros, const fastdiv_t dtx,
                                                          const fastdiv_t dty, const fastdiv_t dpwp, const unsigned out_bytes,
                                                          const int ring, const int xmap, const unsigned in_bytes) {
    ANAT_BEGIN();
    constexpr int WPX = TH / 4;  // tile rows (16-pixel subtiles) per wave
    constexpr int WOC = BN / 16; // every wave covers all BN channels of its rows
    constexpr int NST = WPX;     // buffer stores per wave and tile
    extern __shared__ __attribute__((aligned(16))) int8_t dynlds[];
    uint8_t *slut = (uint8_t *)dynlds; // LDS byte address 0 (requant_pack LUT0)
    lds_base_must_be_zero(dynlds);
    const int nks = k64 / BK;
    constexpr int LB = LUTB + (PRE ? 512 : 0);             // PRE: the 1x1's table behind the main one
    int *dutab = (int *)(dynlds + LB);                     // [nks][4] unit offsets of the K chunks
    int8_t *wl = dynlds + LB + ((nks * 16 + 255) & ~255);   // [nks][BN][64], swizzled like the ring tiles
    const int patch_bytes = nblk * 1024;                   // whole 1 KB blocks (one wave-instruction of LDS-DMA each)
    int8_t *w1l = wl + nks * BN * BK;                      // PRE: [in_c][64] weights of the 1x1
    int8_t *patch0 = w1l + (PRE ? p.in_c * BK : 0);
    int8_t *tpatch = patch0 + ring * patch_bytes;          // PRE: the patch of t

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int oc0 = blockIdx.y * BN;
    const int C = p.in_c, lgc = 31 - __builtin_clz((unsigned)C), cpp = C >> 4, lgcpp = lgc - 4;
    const int s = p.stride_w;
    const unsigned M = C >= 128 ? 6u : (C >= 64 ? 2u : 0u);

    v4i bias[WOC];
#pragma unroll
    for (int q = 0; q < WOC; q++) bias[q] = p.bias ? *(const v4i *)(p.bias + oc0 + q * 16 + (lane >> 4) * 4) : (v4i){0, 0, 0, 0};
    constexpr int W1MAX = 4;
    v4i pbias[PRE ? W1MAX : 1];
    if (PRE) {
#pragma unroll
        for (int q = 0; q < W1MAX; q++) pbias[q] = q < (C >> 4) ? *(const v4i *)(p.pre_bias + q * 16 + (lane >> 4) * 4) : (v4i){0, 0, 0, 0};
    }
    if (HAS_LUT) {
        if (p.lut2) { if (tid < 128) ((uint32_t *)slut)[tid] = ((const uint32_t *)p.lut2)[tid]; }
        else if (tid < 64) ((uint32_t *)slut)[tid] = ((const uint32_t *)p.lut)[tid];
    }
    if (PRE && tid >= 128) ((uint32_t *)slut)[tid] = ((const uint32_t *)p.pre_lut2)[tid - 128]; // LDS 512..1023
    // K chunk table: chunk (ks, f) -> kernel row ky, column kx, channel chunk
    const int rowbytes = p.kw * C, kbytes = p.kh * rowbytes;
    for (int i = tid; i < nks * 4; i += NTHREADS) {
        const int kb = i * 16;
        int du = 0; // K padding meets zero weights: any valid address
        if (kb < kbytes) {
            const int ky = kb / rowbytes, rem = kb - ky * rowbytes, kx = rem >> lgc, cc = (rem & (C - 1)) >> 4;
            const int dp = ky * PWP + (s == 2 ? (kx >> 1) + (kx & 1) * PWH : kx);
            du = dp * cpp + cc;
        }
        dutab[i] = du;
    }
    // Every load the compiler counts (bias, tables) completes HERE, before the first LDS-DMA is issued: a register still
    // pending on the compiler's scoreboard inside the tile loop would make it wait with vmcnt(0), i.e. for every patch in
    // flight (the round-2 form of this kernel did exactly that before the first MFMA of every tile).
    __syncthreads();
    // resident weights: rows oc0 .. oc0+BN-1, every K step (LDS-DMA, source-side swizzle as in the ring kernels)
    {
        const int schunk = (lane & 3) ^ (((lane >> 4) & 1) << 1);
        for (int i = wv; i < nks * (BN / 16); i += 4) {
            const int ks = i / (BN / 16), g = i - ks * (BN / 16);
            glds16(p.w + (size_t)(oc0 + g * 16 + (lane >> 2)) * k64 + ks * BK + schunk * 16, wl + (ks * BN + g * 16) * BK);
        }
        if (PRE)
            for (int g = wv; g < C / 16; g += 4) glds16(p.pre_w + (size_t)(g * 16 + (lane >> 2)) * BK + schunk * 16, w1l + g * 16 * BK);
    }
    // this lane's units of the patch DMA: instruction n of wave wv fills the 1 KB block n*4 + wv = physical units
    // (n*4+wv)*64 + lane; blocks at or beyond nblk do not exist (that instruction is not issued)
    int uoff[PT_NIMAX], upos[PT_NIMAX]; // byte offset from the tile's first input pixel; (py << 16) | px, or -1
#pragma unroll
    for (int n = 0; n < PT_NIMAX; n++) {
        uoff[n] = 0;
        upos[n] = -1;
        if (n * 4 + wv < nblk) {
            const unsigned phys = (unsigned)((n * 4 + wv) * 64 + lane);
            const unsigned U = phys ^ ((phys >> 3) & M);
            const unsigned pp = U >> lgcpp, cc = U & (unsigned)(cpp - 1);
            const unsigned py = fdiv(pp, dpwp), col = pp - py * (unsigned)PWP;
            const int px = s == 2 ? ((int)col < PWH ? 2 * (int)col : 2 * ((int)col - PWH) + 1) : (int)col;
            if ((int)py < PH && px < PW) {
                uoff[n] = ((int)py * p.in_w + px) * C + (int)cc * 16;
                upos[n] = ((int)py << 16) | px;
            }
        }
    }
    const int npw = nblk > wv ? (nblk - wv + 3) >> 2 : 0; // LDS-DMA instructions THIS wave issues per patch
    // xmap (the grid's x extent is a multiple of 8): workgroup ids go round-robin over the 8 XCDs, so XCD x is given the
    // x-th eighth of the tile list and walks it in order: neighbouring tiles (shared halo rows and columns) meet in
    // ONE L2.  Tile id t of a workgroup = 8 * (position in its XCD's range) + xcd.
    const unsigned xcd = blockIdx.x & 7u;
    const unsigned xstart = xmap ? (unsigned)(((unsigned long long)ntiles_all * xcd) >> 3) : 0u;
    const unsigned xend = xmap ? (unsigned)(((unsigned long long)ntiles_all * (xcd + 1u)) >> 3) : 0u;
    const unsigned ntiles = xmap ? (xend - xstart) * 8u + xcd : ntiles_all;
    auto tile_xy = [&](unsigned t, int &tx, int &ty, unsigned &f) {
        const unsigned j = xmap ? xstart + (t >> 3) : t, q = fdiv(j, dtx);
        tx = (int)(j - q * (unsigned)tiles_x);
        f = fdiv(q, dty);
        ty = (int)(q - f * (unsigned)tiles_y);
    };
    // INTERIOR tiles (every patch pixel inside the image: most tiles of a wide map) take the buffer-addressed form: the
    // per-lane offset of a unit never changes (uoff), the tile goes into the instruction's SCALAR offset -- no vector
    // arithmetic per LDS-DMA instruction at all.  (Stamps: the general form below, ~25 instructions per DMA with two bounds
    // tests and a 64-bit select, cost a wave ~1800 cycles per tile in issue alone.)  Needs 31-bit offsets (in_bytes != 0).
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, (int)in_bytes, 0x00020000);
    auto issue_patch = [&](unsigned t, int8_t *dst) {
#ifdef PATCH_ABL // timing-only build (tools/stamps_build.sh patchabl 1): no patch is fetched -- what the layer costs with its input already in LDS
        return;
#endif
        int tx, ty;
        unsigned f;
        tile_xy(t, tx, ty, f);
        const int iy0 = ty * TH * s - p.pad_top, ix0 = tx * PT_TW * s - p.pad_left;
        if (in_bytes != 0u && iy0 >= 0 && ix0 >= 0 && iy0 + PH <= p.in_h && ix0 + PW <= p.in_w) {
            const int sbase = (int)(f * (unsigned)p.in_stride) + (iy0 * p.in_w + ix0) * C;
#pragma unroll
            for (int n = 0; n < PT_NIMAX; n++)
                if (n * 4 + wv < nblk) blds16(xrs, uoff[n], sbase, dst + (n * 4 + wv) * 1024); // pad lanes (uoff 0): bytes never read
            return;
        }
        const int8_t *base = p.in + (size_t)f * p.in_stride + ((long)iy0 * p.in_w + ix0) * C;
#pragma unroll
        for (int n = 0; n < PT_NIMAX; n++)
            if (n * 4 + wv < nblk) {
                const int py = upos[n] >> 16, px = upos[n] & 0xffff;
                const bool ok = upos[n] >= 0 && (unsigned)(iy0 + py) < (unsigned)p.in_h && (unsigned)(ix0 + px) < (unsigned)p.in_w;
                glds16(ok ? base + uoff[n] : zeros, dst + (n * 4 + wv) * 1024);
            }
    };

    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)out_bytes, 0x00020000);
    // the residual operand of a fused Add has the output's layout: same offsets, same extent
    const bool has_add = p.add != nullptr;
    // residual staging: per wave WPX rows of 64 lanes x 16 bytes (BN 64) or 2 x 64 lanes x 4 bytes (BN 32), behind the patches
    constexpr int RROW = WOC == 4 ? 1024 : 512;
    constexpr int NRI = WOC == 4 ? WPX : 2 * WPX;   // LDS-DMA instructions per wave and tile for it
    int8_t *rstage = patch0 + (ring + (PRE ? 1 : 0)) * patch_bytes + wv * (WPX * RROW);
    const int NR = has_add ? NRI : 0;
    const int frow = lane & 15, fchunk = lane >> 4;
    const int chan = (lane >> 4) * (4 * WOC);
    const int pstride = p.out_pix_stride ? p.out_pix_stride : p.out_c;
    const int lo = p.relu ? 0 : -128;
    const uint8_t *lut128 = slut + 128;
    int ubase[WPX]; // 16-byte unit of (tile row, column frow), tap (0,0), channel 0
#pragma unroll
    for (int u = 0; u < WPX; u++) ubase[u] = ((wv * WPX + u) * s * PWP + frow) * cpp;

    // ---- the ring.  D = ring - 1 patches are in flight behind the one being computed.  yg[k] = vector-memory operations
    // this wave has issued AFTER the k-th oldest outstanding patch (loads, LDS-DMA and stores retire in order against one
    // counter on gfx9, so "patch landed" == "at most yg outstanding").
    const unsigned G = gridDim.x;
    const int D = ring - 1;
    unsigned t = blockIdx.x;
    int yg0 = 0, yg1 = 0, yg2 = 0;
    {
        int issued = 0;
        for (int d = 0; d < D; d++)
            if (t + (unsigned)d * G < ntiles) {
                issue_patch(t + (unsigned)d * G, patch0 + d * patch_bytes);
                issued++;
            }
        // younger than the d-th patch: the patches issued after it
        yg0 = issued > 1 ? (issued - 1) * npw : 0;
        yg1 = issued > 2 ? (issued - 2) * npw : 0;
        yg2 = 0;
    }
#ifdef PATCH_STAMPS
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, sum_wait = 0, sum_issue = 0, sum_k = 0, sum_epi = 0, sum_tiles = 0;
#endif
    int cur = 0;       // ring slot of the tile being computed
    int pn_last = 0;   // LDS-DMA instructions issued after this tile's residual request
    bool first = true;
    ANAT_NOW(1);
    for (; t < ntiles; t += G) {
        STAMP(st0);
        int tx, ty;
        unsigned f;
        tile_xy(t, tx, ty, f);
        // output offsets of this lane's pixels; with a fused residual Add the other operand (same layout) is requested
        // now, ahead of everything else of this tile, by loads the compiler does not count (a counted load would be waited
        // for with vmcnt(0): it cannot know how many LDS-DMA instructions follow)
        int voffs[WPX];
        uint32_t xw[WPX][WOC];
#pragma unroll
        for (int u = 0; u < WPX; u++) {
            const int oy = ty * TH + wv * WPX + u, ox = tx * PT_TW + frow;
            const unsigned off = f * (unsigned)p.out_stride + (unsigned)(oy * p.out_w + ox) * (unsigned)pstride +
                                 (unsigned)(p.out_ch_off + oc0 + chan);
            const bool ok = oy < p.out_h && ox < p.out_w && oc0 + chan < p.out_c;
            voffs[u] = ok ? (int)off : -1; // out of range for the buffer unit: loads return 0, stores are dropped
#pragma unroll
            for (int q = 0; q < WOC; q++) xw[u][q] = 0;
        }
        if (has_add) {
#pragma unroll
            for (int u = 0; u < WPX; u++) {
                const int8_t *src = voffs[u] >= 0 ? p.add + (unsigned)voffs[u] : zeros;
                if (WOC == 4) {
                    glds16(src, rstage + u * RROW);
                } else { // 8 bytes per lane: two dword pieces, each 64 lanes x 4 bytes
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(rstage + u * RROW), 4, 0, 0);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 4),
                                                     (__attribute__((address_space(3))) void *)(rstage + u * RROW + 256), 4, 0, 0);
                }
            }
        }
        if (D > 0) {
            // this tile's patch (and, the first time, the weights: older still) has landed when at most the operations
            // issued after it are outstanding
            wait_vmcnt_upto<63>(yg0 + NR);
            barrier_lds(); // ... for every wave; and every wave is past its reads of the slot refilled next
            STAMP(st1);
            const unsigned tn = t + (unsigned)D * G;
            int pn = 0;
            if (tn < ntiles) {
                int slot = cur + D;
                slot = slot >= ring ? slot - ring : slot;
                issue_patch(tn, patch0 + slot * patch_bytes);
                pn = npw;
            }
            // book-keeping for the NEXT tile: its patch is the second oldest now
            const int after = NR + pn + NST;
            yg0 = yg1 + after;
            yg1 = yg2 + after;
            yg2 = NST; // the patch issued just now (if any): only this tile's stores follow it
            if (D == 1) yg0 = NST;
            if (D == 2) yg1 = NST;
            pn_last = pn;
        } else {
            // one patch buffer: every wave must be past its reads of the previous tile before the buffer is refilled
            if (!first) barrier_lds();
            issue_patch(t, patch0);
            wait_vmcnt<0>();
            barrier_lds();
        }
        first = false;
        STAMP(st2);
        const int8_t *patch = patch0 + cur * patch_bytes;
        if (PRE) {
            // stage 1: t = SiLU(requant(W1 x + b1)) for every pixel of the patch, 16 flat pixels per MFMA column block;
            // lane (i, g) ends with the 4 * WOC1 consecutive channels g * 4 * WOC1 .. of pixel i (row order of the packer)
            const int WOC1 = C >> 4;
            const int P = PH * PWP, nsub = (P + 15) >> 4;
            const int iy0 = ty * TH * s - p.pad_top, ix0 = tx * PT_TW * s - p.pad_left;
            for (int sub = wv; sub < nsub; sub += 4) {
                int pp = sub * 16 + frow;
                const bool live = pp < P;
                pp = live ? pp : P - 1;
                const unsigned Ur = (unsigned)(pp * cpp + (fchunk & (cpp - 1))); // in_c 32: chunks 2, 3 meet zero weights
                const v4i xb1 = *(const v4i *)(patch + ((Ur ^ ((Ur >> 3) & M)) << 4));
                const unsigned py = fdiv((unsigned)pp, dpwp), px = (unsigned)pp - py * (unsigned)PWP;
                const bool inimg = live && (int)py < PH && (int)px < PW && (unsigned)(iy0 + (int)py) < (unsigned)p.in_h &&
                                   (unsigned)(ix0 + (int)px) < (unsigned)p.in_w;
                uint32_t pk1[W1MAX];
                if (WOC1 == 4) {
                    int a1[16];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const v4i wa = *(const v4i *)(w1l + lds_off(q * 16 + frow, fchunk));
                        const v4i r = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, xb1, pbias[PRE ? q : 0], 0, 0, 0);
#pragma unroll
                        for (int e = 0; e < 4; e++) a1[q * 4 + e] = r[e];
                    }
                    uint32_t pk4[4];
                    requant_pack_pre<16>(a1, p.pre_cs, pk4);
#pragma unroll
                    for (int e = 0; e < 4; e++) pk1[e] = inimg ? pk4[e] : 0u;
                    const unsigned Uw = (unsigned)(pp * cpp + (lane >> 4));
                    if (live) *(v4i *)(tpatch + ((Uw ^ ((Uw >> 3) & M)) << 4)) = (v4i){(int)pk1[0], (int)pk1[1], (int)pk1[2], (int)pk1[3]};
                } else {
                    int a1[8];
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const v4i wa = *(const v4i *)(w1l + lds_off(q * 16 + frow, fchunk));
                        const v4i r = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, xb1, pbias[PRE ? q : 0], 0, 0, 0);
#pragma unroll
                        for (int e = 0; e < 4; e++) a1[q * 4 + e] = r[e];
                    }
                    uint32_t pk2[2];
                    requant_pack_pre<8>(a1, p.pre_cs, pk2);
                    const int g4 = lane >> 4; // channels g4 * 8 .. + 7: unit g4 >> 1, byte (g4 & 1) * 8
                    const unsigned Uw = (unsigned)(pp * cpp + (g4 >> 1));
                    if (live)
                        *(uint2 *)(tpatch + ((Uw ^ ((Uw >> 3) & M)) << 4) + (g4 & 1) * 8) = make_uint2(inimg ? pk2[0] : 0u, inimg ? pk2[1] : 0u);
                }
            }
            barrier_lds(); // t complete: the K loop below reads it in place of x
            patch = tpatch;
        }
        // K loop, software-pipelined by hand: the fragments of step k+1 (and the table entry of step k+2) are requested
        // before the MFMAs of step k are issued.  Stamps of the round-2 loop (one step = table read -> address arithmetic ->
        // 8 ds_read_b128 -> wait -> 16 MFMAs, in series) showed ~620 cycles per step for 256 cycles of matrix work: with two
        // waves per SIMD (196 registers) nothing else hides a wave's LDS latency.
        v4i acc[WOC][WPX];
        v4i xa[WPX], wa[WOC], xb[WPX], wb[WOC];
        auto load_frags = [&](int ks, int du, v4i (&xf)[WPX], v4i (&wf)[WOC]) {
#pragma unroll
            for (int u = 0; u < WPX; u++) {
                const unsigned U = (unsigned)(ubase[u] + du);
                xf[u] = *(const v4i *)(patch + ((U ^ ((U >> 3) & M)) << 4));
            }
            const int8_t *ws = wl + ks * BN * BK;
#pragma unroll
            for (int q = 0; q < WOC; q++) wf[q] = *(const v4i *)(ws + lds_off(q * 16 + frow, fchunk));
        };
        auto mfma_step = [&](const v4i (&xf)[WPX], const v4i (&wf)[WOC]) {
#pragma unroll
            for (int q = 0; q < WOC; q++)
#pragma unroll
                for (int u = 0; u < WPX; u++) acc[q][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[q], xf[u], acc[q][u], 0, 0, 0);
        };
        // (every load below is unconditional -- indices past the last step are clamped to it and the bytes ignored -- so
        // that the compiler can count its LDS operations: behind a branch it falls back to lgkmcnt(0), which would wait
        // for the reads just issued for the NEXT step before this step's MFMAs)
        const int last = nks - 1;
        auto clampk = [&](int k) { return k < last ? k : last; };
        load_frags(0, dutab[fchunk], xa, wa);
        load_frags(clampk(1), dutab[clampk(1) * 4 + fchunk], xb, wb);
        int du_n = dutab[clampk(2) * 4 + fchunk];
        // step 0 takes the bias as its C operand: the accumulators start there
#pragma unroll
        for (int q = 0; q < WOC; q++)
#pragma unroll
            for (int u = 0; u < WPX; u++) acc[q][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[q], xa[u], bias[q], 0, 0, 0);
        int ks = 1;
        for (; ks + 1 < nks; ks += 2) { // fragments of step ks sit in (xb, wb); du_n = table entry of step ks + 1
            // (sched_barrier: left alone, the scheduler sinks the reads between the MFMAs that need them and waits for
            // each group with lgkmcnt(0))
            load_frags(ks + 1, du_n, xa, wa);
            du_n = dutab[clampk(ks + 2) * 4 + fchunk];
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(xb, wb);
            __builtin_amdgcn_sched_barrier(0);
            load_frags(clampk(ks + 2), du_n, xb, wb);
            du_n = dutab[clampk(ks + 3) * 4 + fchunk];
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(xa, wa);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ks < nks) mfma_step(xb, wb); // an even number of steps: the last one
        STAMP(st3);
        if (D > 0) cur = cur + 1 == ring ? 0 : cur + 1;
        if (has_add) {
            // the residual rows of this wave have landed when only the patch issued after them is outstanding (D == 0: the
            // wait for the patch was a full one); the same lane that requested a piece reads it back
            if (D > 0) wait_vmcnt_upto<15>(pn_last);
#pragma unroll
            for (int u = 0; u < WPX; u++) {
                if (WOC == 4) {
                    const v4i r = *(const v4i *)(rstage + u * RROW + lane * 16);
                    xw[u][0] = (uint32_t)r[0]; xw[u][1] = (uint32_t)r[1]; xw[u][WOC > 2 ? 2 : 0] = (uint32_t)r[2]; xw[u][WOC > 3 ? 3 : 0] = (uint32_t)r[3];
                } else {
                    xw[u][0] = *(const uint32_t *)(rstage + u * RROW + lane * 4);
                    xw[u][WOC > 1 ? 1 : 0] = *(const uint32_t *)(rstage + u * RROW + 256 + lane * 4);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < WPX; u++) {
            uint32_t pk[WOC];
            int a[WOC * 4];
#pragma unroll
            for (int q = 0; q < WOC; q++)
#pragma unroll
                for (int r = 0; r < 4; r++) a[q * 4 + r] = acc[q][u][r];
            const bool fast = HAS_LUT && p.lut2 != nullptr;
            if (has_add) {
                const add_args_t ga = {p.add_s_conv, p.add_s_other, p.add_inv};
                if (fast) requant_pack<WOC * 4, HAS_LUT, true, true, true, true>(a, p.cs, lo, lut128, pk, xw[u], &ga);
                else requant_pack<WOC * 4, HAS_LUT, true, true, true>(a, p.cs, lo, lut128, pk, xw[u], &ga);
            } else {
                if (fast) requant_pack<WOC * 4, HAS_LUT, true, true, false, true>(a, p.cs, lo, lut128, pk);
                else requant_pack<WOC * 4, HAS_LUT, true, true>(a, p.cs, lo, lut128, pk);
            }
            const int voff = voffs[u];
            if (WOC == 4)
                __builtin_amdgcn_raw_buffer_store_b128((v4i){(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]}, orsrc, voff, 0, 0);
            else if (WOC == 2)
                __builtin_amdgcn_raw_buffer_store_b64((v2i){(int)pk[0], (int)pk[WOC > 1 ? 1 : 0]}, orsrc, voff, 0, 0);
        }
#ifdef PATCH_STAMPS
        STAMP(st4);
        if (D == 0) st1 = st2; // one buffer: issue and wait are one segment
        sum_wait += st1 - st0; sum_issue += st2 - st1; sum_k += st3 - st2; sum_epi += st4 - st3; sum_tiles += 1;
#endif
    }
    ANAT_NOW(2);
    ANAT_END(p);
#ifdef PATCH_STAMPS
    if (lane == 0) {
        atomicAdd(&patch_stamp_sums[0], sum_wait); atomicAdd(&patch_stamp_sums[1], sum_issue); atomicAdd(&patch_stamp_sums[2], sum_k);
        atomicAdd(&patch_stamp_sums[3], sum_epi); atomicAdd(&patch_stamp_sums[4], sum_tiles);
    }
#endif
}

ANAT_SETTER(mhip_anatomy_set_patch)
// ---- patch-staged kernel: geometry, eligibility, launch
struct patch_geom_t {
    int bn, tiles_x, tiles_y, PH, PW, PWP, PWH, nblk, nks, ring;
    size_t lds;
};
static inline const tune_t &tune() { return conv_i8_tune_state(); }
// LDS budget of one workgroup.  Measured (tools/layer_time.py, yolov5s shapes, batch 256): workgroups per CU matter
// more than ring depth -- three workgroups with two patch buffers beat two with four (3x3 32->32 @160: 194 vs 225 us) -- and
// beyond one patch in flight per workgroup nothing is gained.  The 32-channel-tile instantiations fit three waves per SIMD
// (<= 168 registers): they get a third of the CU; the 64-channel tiles (200+ registers: two waves per SIMD) half of it.
static size_t patch_lds_budget(int bn) {
    const int kb = tune().patch_lds_kb;
    if (kb >= 16 && kb <= 160 && kb != 80) return (size_t)kb * 1024; // forced (experiments / tests)
    return (size_t)(bn == 32 ? 53 : 80) * 1024;
}
static bool patch_geom(const mhip_conv_i8_t *p, int th, patch_geom_t *g) {
    const bool direct = conv_i8_direct_rows(p);
    const int C = p->in_c, s = p->stride_w;
    // in_c == 16 (round 6: the second layer of the yolov5n models, 3 x 3 stride 2 on 320 x 320 x 16 -- one 16-byte unit per pixel, no swizzle
    // needed: 16 consecutive pixels are 16 consecutive units); MARS_HIP_PATCH_NO_C16 sends it back to the implicit-GEMM form (A / B)
    static const bool no_c16 = getenv("MARS_HIP_PATCH_NO_C16") != nullptr;
    if (!direct || !p->safe || (C != 16 && C != 32 && C != 64 && C != 128) || (C == 16 && (no_c16 || p->pre_w)) || (s != 1 && s != 2) || p->stride_h != s ||
        p->kh > 7 || p->kw > 7 || p->kh * p->kw < 2 || p->row_pad != p->kw * C || persist_out_bytes(p) > 0x7fffffffL)
        return false;
    const int k64 = (p->kh * p->row_pad + BK - 1) / BK * BK;
    g->nks = k64 / BK;
    g->bn = p->oc_pad % 64 == 0 ? 64 : 32;
    g->tiles_x = (p->out_w + PT_TW - 1) / PT_TW;
    g->tiles_y = (p->out_h + th - 1) / th;
    // mostly full tiles only: a tile computes th x 16 pixels whether the image has them or not
    if ((double)p->out_h * p->out_w < (C == 128 ? 0.80 : 0.85) * (double)g->tiles_x * PT_TW * g->tiles_y * th) return false;
    g->PH = (th - 1) * s + p->kh;
    g->PW = (PT_TW - 1) * s + p->kw;
    g->PWH = s == 2 ? (g->PW + 1) / 2 : 0;
    g->PWP = s == 2 ? 2 * g->PWH : g->PW;
    const long units = (long)g->PH * g->PWP * (C / 16);
    g->nblk = (int)((units + 63) / 64); // the unit swizzle permutes inside 64-unit blocks: whole blocks
    if (g->nblk > 4 * PT_NIMAX) return false;
    const bool pre = p->pre_w != nullptr; // + the 1x1's table, its weights and the patch of its output
    if (pre && (s != 1 || (C != 32 && C != 64) || !p->pre_bias || !p->pre_lut2 || !p->lut2)) return false;
    const size_t pb = (size_t)g->nblk * 1024;
    const size_t radd = p->add ? (size_t)4 * (th / 4) * (g->bn == 64 ? 1024 : 512) : 0; // residual staging rows of the 4 waves
    const size_t fixed = LUTB + (pre ? 512 + (size_t)C * BK + pb : 0) + (((size_t)g->nks * 16 + 255) & ~(size_t)255) +
                         (size_t)g->nks * g->bn * BK + radd;
    const size_t budget = patch_lds_budget(g->bn);
    if (fixed + pb > budget) return false;
    // as many patch buffers as the budget holds, at most 4 (three patches in flight behind the one being computed);
    // MARS_HIP_PATCH_RING / the "patch_ring" knob caps it (tests force every depth)
    // default: at most two buffers (the kernel takes up to 4: deeper rings measured equal or slower, see patch_lds_budget);
    // the "patch_ring" knob (tests, experiments) sets the cap, 1..4
    const int fit = (int)((budget - fixed) / pb);
    const int cap = tune().patch_ring > 0 ? (tune().patch_ring > 4 ? 4 : tune().patch_ring) : 2;
    int ring = fit < cap ? fit : cap;
    g->ring = ring;
    g->lds = fixed + (size_t)ring * pb;
    if ((long)g->tiles_x * g->tiles_y * p->frames > 0x7fffffffL) return false;
    return true;
}

template <int TH, int BN, bool HAS_LUT, bool PRE = false>
static int launch_patch_t(const mhip_conv_i8_t *p, int k64, const patch_geom_t &g) {
    auto kern = conv_i8_patch<TH, BN, HAS_LUT, PRE>;
    // workgroups the device holds at once at THIS layer's LDS size (small patches fit 3-4 per CU), cached per size
    static int cus = 0;
    static size_t slots_lds[8];
    static int slots_n[8], nslots = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return mhip_check(hipErrorUnknown, "conv_i8_patch occupancy query");
        cus = prop.multiProcessorCount;
    }
    int slots = 0;
    for (int i = 0; i < nslots; i++)
        if (slots_lds[i] == g.lds) slots = slots_n[i];
    if (!slots) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, NTHREADS, g.lds) != hipSuccess)
            return mhip_check(hipErrorUnknown, "conv_i8_patch occupancy query");
        slots = (occ > 0 ? occ : 1) * cus;
        if (nslots < 8) { slots_lds[nslots] = g.lds; slots_n[nslots++] = slots; }
    }
    const unsigned ntiles = (unsigned)((long)g.tiles_x * g.tiles_y * p->frames);
    const unsigned noc = (unsigned)(p->oc_pad / BN);
    unsigned gx = (unsigned)(tune().persist_slots > 0 ? tune().persist_slots : slots) / noc;
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    const int xmap = gx >= 8 && ntiles < 0x0fffffffu; // ids reach 8 x the longest range
    if (xmap) gx &= ~7u;
    {
        static const char *dbg = getenv("MARS_HIP_PATCH_DEBUG");
        static size_t last_lds = 0;
        static int last_ring = 0;
        if (dbg && (last_lds != g.lds || last_ring != g.ring)) {
            fprintf(stderr, "Mars: conv_i8_patch<%d,%d> ring %d, patch %d KB, LDS %zu B, %d workgroups per CU, grid %u x %u\n", TH, BN, g.ring,
                    g.nblk, g.lds, slots / cus, gx, noc);
            last_lds = g.lds;
            last_ring = g.ring;
        }
    }
    hipLaunchKernelGGL(kern, dim3(gx, noc), dim3(NTHREADS), g.lds, mhip_stream_native(), *p, k64,
                       g.tiles_x, g.tiles_y, ntiles, g.PH, g.PW, g.PWP, g.PWH, g.nblk, (const int8_t *)mhip_zero_page(),
                       make_fastdiv((unsigned)g.tiles_x), make_fastdiv((unsigned)g.tiles_y), make_fastdiv((unsigned)g.PWP),
                       (unsigned)persist_out_bytes(p), g.ring, xmap,
                       in_extent_bytes(p) <= 0x7fffffffL ? (unsigned)in_extent_bytes(p) : 0u);
    return mhip_check(hipGetLastError(), "conv_i8_patch launch");
}

bool conv_i8_patch_ok(const mhip_conv_i8_t *p, int th, int *ring) {
    patch_geom_t g;
    if (!patch_geom(p, th, &g)) return false;
    if (ring) *ring = g.ring;
    return true;
}

int conv_i8_launch_patch(const mhip_conv_i8_t *p, int k64, int th) {
    patch_geom_t g;
    if (!patch_geom(p, th, &g)) return -1;
#define PATCH(T, B)                                                                       \
    (p->pre_w ? launch_patch_t<T, B, true, true>(p, k64, g)                               \
              : (p->lut ? launch_patch_t<T, B, true>(p, k64, g) : launch_patch_t<T, B, false>(p, k64, g)))
    if (th == 16) return g.bn == 64 ? PATCH(16, 64) : PATCH(16, 32);
    if (th == 8) return g.bn == 64 ? PATCH(8, 64) : PATCH(8, 32);
    return g.bn == 64 ? PATCH(4, 64) : PATCH(4, 32);
#undef PATCH
}

// fused bottleneck (pre_* fields): only the patch-staged kernel evaluates it; some tile height must fit
int conv_i8_pre_tile_rows(const mhip_conv_i8_t *p) {
    patch_geom_t g;
    for (int th : {16, 8, 4})
        if (patch_geom(p, th, &g) && (th == 4 || g.ring >= 2)) return th;
    return 0;
}
extern "C" int mhip_conv_i8_pre_ok(const mhip_conv_i8_t *p) {
    if (!p || !p->pre_w || p->nseg > 1 || p->out_nchw) return 0;
    return conv_i8_pre_tile_rows(p) != 0;
}
