// conv_i8_stem.hip -- int8 convolutions with at most 4 input channels (the RGB stem) for gfx950.
// Replaces reference src/mars/mxu_conv.c:713-757 for in_c <= 4; arithmetic contract and epilogue: conv_i8_common.hpp.
#include "conv_i8_common.hpp"

// ---------------------------------------------------------------------------------
// small-channel kernel (in_c <= 4, kw <= 8: the RGB stem).  The input patch of an
// 8x16 output tile is staged ONCE in LDS with every pixel widened to 4 bytes, so a
// kernel row of a pixel is 32 contiguous LDS bytes (kw*4 used, the rest meets zero
// weights) and one MFMA K step covers two kernel rows.  Weights ([oc][kh][32]) stay in
// LDS for the lifetime of the (persistent) workgroup.  Input bytes are read once.
#define SC_TH 16
#define SC_TW 16
#define SC_BP (SC_TH * SC_TW)
// (WOC == 1, the general form: 97 VGPRs, one over the 96 of five waves per SIMD -- the hint brings it to 91 without a spill, +1.3 % on
//  yolov5n_int8.mars; six waves (80) spill 9 registers and lose 2 %; the same hint on conv_i8_persist<128, 32> -- 83 -> 80 -- measured nothing)
template <int WOC, bool HOT = false>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(WOC == 1 ? 5 : 1, 8))) void conv_i8_smallc(const mhip_conv_i8_t p, const int k64, const int tiles_x,
                                                           const int tiles_y, const unsigned ntiles_all, const int PH,
                                                           const int PW, const int PWp, const fastdiv_t dhw,
                                                           const fastdiv_t dtx, const fastdiv_t dty, const fastdiv_t dgpr,
                                                           const int tile_bytes) {
    ANAT_BEGIN();
    constexpr int BN = WOC * 16;
    constexpr int WPX = SC_TH / 4; // tile rows (= pixel subtiles of 16) per wave
    extern __shared__ __attribute__((aligned(16))) int8_t dyn[];
    uint8_t *slut = (uint8_t *)dyn;                  // LDS byte address 0 (no static LDS here: requant_pack LUT0)
    long *rowoff = (long *)(dyn + LUTB);             // [256]
    int8_t *wl = dyn + LUTB + SC_BP * 8;             // [k64/64][BN][64], rows swizzled like the ring tiles (lds_off)
    int8_t *patch0 = wl + BN * k64;                  // 2 x [(PH+1)][PWp] dwords (double buffer)
    const int patch_bytes = ((PH + 1) * PWp * 4 + 15) & ~15;
    int8_t *tile = patch0 + 2 * patch_bytes;         // [256][BN+OPAD]
    v4i *sbias = (v4i *)(tile + tile_bytes);          // [BN / 4]; tile_bytes = 0 when the rows are stored straight from
                                                      // registers (NHWC, 16-byte aligned): 9 KB less, so that 4 of these workgroups
                                                      // still share a CU with the 36 KB NMS workgroup of the previous batch
    lds_base_must_be_zero(dyn);

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int hw = p.out_h * p.out_w;
    if (p.lut2) { if (tid < 128) ((uint32_t *)slut)[tid] = ((const uint32_t *)p.lut2)[tid]; }
    else if (p.lut && tid < 64) ((uint32_t *)slut)[tid] = ((const uint32_t *)p.lut)[tid];
    // WOC == 1 (out_c <= 16 in a 32-row image): the host packed rows and bias in the two-subtile order (conv_i8_common.hpp: MFMA row 4 g + r
    // of subtile s carries channel 8 g + 4 s + r); this form wants channel c in row c: it reads packed row 16 s + 4 g + r of channel c
    auto packed_row = [](int c) { return WOC == 1 ? ((c >> 2) & 1) * 16 + (c >> 3) * 4 + (c & 3) : c; };
    for (int i = tid; i < BN * (k64 / 16); i += NTHREADS) {
        const int row = i / (k64 / 16), c = i - row * (k64 / 16);
        *(v4i *)(wl + (c >> 2) * (BN * BK) + lds_off(row, c & 3)) = *(const v4i *)(p.w + (size_t)packed_row(row) * k64 + c * 16);
    }
    for (int i = tid; i < 2 * patch_bytes / 4; i += NTHREADS) ((uint32_t *)patch0)[i] = 0;
    if (tid < BN) ((int *)sbias)[tid] = p.bias ? p.bias[packed_row(tid)] : 0;

    // one staging unit = 4 consecutive patch pixels of one row -> one 16-byte LDS store.
    // in_c == 3: the 12 source bytes come from ONE unaligned 16-byte global load (gfx950 serves
    // global accesses at any byte alignment) when all 4 pixels are inside the image.
    const int gpr = (PW + 3) >> 2;           // units per patch row
    const int nunits = PH * gpr;             // host guarantees nunits <= 2 * NTHREADS
    // this thread's (at most 2) units never change: patch row r, pixel group g
    // (row << 8 | group) in one register each: the kernel sits at 112 VGPRs, one allocation granule below 120, so that
    // 4 of its waves still fit beside a 64-register wave of the detection tail (measured: -2 % per batch at 115)
    int urg[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const unsigned u = (unsigned)(tid + j * NTHREADS);
        const int r = (int)fdiv(u, dgpr);
        urg[j] = (r << 8) | ((int)u - r * gpr);
    }
    v4i pre[2];
    int shf[2]; // fast path: column shift of the loaded pixels (PRE_ZERO: nothing of this unit is inside the image)
    constexpr int PRE_ZERO = 8, PRE_DONE = -100;
    // (round 6: 4-byte pixels -- an NCHW-tagged graph's 3-plane input relaid to [HW][4] -- take the same clamped 16-byte loads: a unit is
    // 16 contiguous bytes there; they used to go byte by byte under bounds tests, 0.85 ms for the shipped yolov5n_int8.mars stem)
    const int fastpb = HOT ? 3 : (p.in_w >= 4 ? (p.in_c == 3 ? 3 : (p.in_c == 4 ? 4 : 0)) : 0); // bytes per pixel of the fast form, 0 = the gather
    const bool fast3 = fastpb != 0;
    // planar input (in_planar planes [c][H][W]: an NCHW-tagged graph's input, host: in_c == 4, in_w >= 4): a unit is 4 consecutive pixels = ONE dword
    // per plane, interleaved into 4-byte pixels at commit -- the fast form with three (four) loads instead of one, no relayout launch in front
    const int planar = HOT ? 0 : p.in_planar;
    const long plane = (long)p.in_h * p.in_w;
    // Workgroup ids go round-robin over the 8 XCDs (the grid is a multiple of 8, so a workgroup's XCD is blockIdx.x & 7
    // for its whole run): XCD x is given the x-th eighth of the tile list and walks it in order, so the workgroups
    // that share patch halos and 128-byte input lines run side by side under ONE L2 (measured: the kernel fetched
    // 4.5x its input when neighbouring tiles sat on different XCDs, 1.4x now).  Tile id t = 8 * (position in the
    // XCD's range) + xcd; ids below `ntiles` are valid.
    const unsigned xcd = blockIdx.x & 7u;
    const unsigned xstart = (unsigned)(((unsigned long long)ntiles_all * xcd) >> 3);
    const unsigned ntiles = ((unsigned)(((unsigned long long)ntiles_all * (xcd + 1u)) >> 3) - xstart) * 8u + xcd;
    auto tile_xy = [&](unsigned t, int &tx, int &ty, unsigned &f) {
        const unsigned j = xstart + (t >> 3), q = fdiv(j, dtx);
        tx = (int)(j - q * (unsigned)tiles_x);
        f = fdiv(q, dty);
        ty = (int)(q - f * (unsigned)tiles_y);
    };
    auto fetch = [&](unsigned t) {
        int tx, ty;
        unsigned f;
        tile_xy(t, tx, ty, f);
        const int8_t *src = p.in + (size_t)f * p.in_stride;
        const int y0 = ty * SC_TH * p.stride_h - p.pad_top, x0 = tx * SC_TW * p.stride_w - p.pad_left;
        if (fast3) {
            // EVERY lane issues its loads unconditionally, from an address clamped into the image, and nothing looks
            // at the bytes before commit(): the loads stay in flight across this tile's MFMAs (a load under a
            // divergent branch is waited for inside the branch).  Units over the left / right edge load the 4 pixels
            // at the clamped column and are shifted into place at commit (zeros move in).
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int iy = y0 + (urg[j] >> 8), ix = x0 + (urg[j] & 255) * 4;
                const int iyc = iy < 0 ? 0 : (iy > p.in_h - 1 ? p.in_h - 1 : iy);
                const int ixc = ix < 0 ? 0 : (ix > p.in_w - 4 ? p.in_w - 4 : ix);
                if (planar) {
                    const int8_t *q = src + (long)iyc * p.in_w + ixc;
                    int w4[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        if (c < planar) __builtin_memcpy(&w4[c], q + c * plane, 4); // unaligned dword (uniform branch: every lane loads)
                    pre[j] = (v4i){w4[0], w4[1], w4[2], w4[3]};
                } else {
                    __builtin_memcpy(&pre[j], src + ((long)iyc * p.in_w + ixc) * fastpb, 16); // unaligned dwordx4, 12 (16) bytes used
                }
                shf[j] = (iy == iyc && tid + j * NTHREADS < nunits) ? ixc - ix : PRE_ZERO;
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            v4i v = {0, 0, 0, 0};
            if (tid + j * NTHREADS < nunits) {
                const int iy = y0 + (urg[j] >> 8), ix = x0 + (urg[j] & 255) * 4;
                if (iy >= 0 && iy < p.in_h) {
                    const int8_t *q = src + ((long)iy * p.in_w + ix) * p.in_c;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        uint32_t w = 0;
                        if (ix + e >= 0 && ix + e < p.in_w)
                            for (int c = 0; c < p.in_c; c++) w |= (uint32_t)(uint8_t)q[e * p.in_c + c] << (8 * c);
                        v[e] = (int)w;
                    }
                }
            }
            pre[j] = v;
            shf[j] = PRE_DONE;
        }
    };
    auto commit = [&](int8_t *patch) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            v4i v = pre[j];
            if (fast3) { // 4 x 3 packed bytes -> 4 pixels widened to a dword each
                const uint32_t d0 = (uint32_t)v[0], d1 = (uint32_t)v[1], d2 = (uint32_t)v[2];
                v4i l = v; // (4-byte pixels: already one dword each)
                if (planar) { // dword c = pixels 0 .. 3 of plane c  ->  dword e = channels 0 .. 3 of pixel e (a 4 x 4 byte transpose)
                    const uint32_t a = (uint32_t)v[0], b = (uint32_t)v[1], c2 = (uint32_t)v[2], d = (uint32_t)v[3];
                    const uint32_t ab_lo = __builtin_amdgcn_perm(b, a, 0x05010400u), ab_hi = __builtin_amdgcn_perm(b, a, 0x07030602u); // a0 b0 a1 b1 | a2 b2 a3 b3
                    const uint32_t cd_lo = __builtin_amdgcn_perm(d, c2, 0x05010400u), cd_hi = __builtin_amdgcn_perm(d, c2, 0x07030602u);
                    l[0] = (int)__builtin_amdgcn_perm(cd_lo, ab_lo, 0x05040100u); // a0 b0 c0 d0
                    l[1] = (int)__builtin_amdgcn_perm(cd_lo, ab_lo, 0x07060302u);
                    l[2] = (int)__builtin_amdgcn_perm(cd_hi, ab_hi, 0x05040100u);
                    l[3] = (int)__builtin_amdgcn_perm(cd_hi, ab_hi, 0x07060302u);
                } else if (HOT || fastpb == 3) {
                    l[0] = (int)(d0 & 0xFFFFFFu);
                    l[1] = (int)(((d0 >> 24) | (d1 << 8)) & 0xFFFFFFu);
                    l[2] = (int)(((d1 >> 16) | (d2 << 16)) & 0xFFFFFFu);
                    l[3] = (int)(d2 >> 8);
                }
                v = l;
                if (shf[j] != 0) { // patch pixel e is loaded pixel e - shift
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int k = e - shf[j];
                        v[e] = k == 0 ? l[0] : (k == 1 ? l[1] : (k == 2 ? l[2] : (k == 3 ? l[3] : 0)));
                    }
                }
            }
            if (tid + j * NTHREADS < nunits) *(v4i *)(patch + ((size_t)(urg[j] >> 8) * PWp + (urg[j] & 255) * 4) * 4) = v;
        }
    };

    // Software pipeline over tiles with a double-buffered patch: the next tile's loads are issued
    // before this tile's MFMAs and committed to the OTHER buffer before this tile's stores, so the
    // (in-order) vmcnt wait for those loads never sits behind freshly issued stores.
    const int nks = k64 / BK;
    // MFMA operand addresses of this lane: B = pixel (row wv*WPX+u, column lane&15), K chunk c = lane>>4 -> kernel row
    // 2*ks + (c>>1), pixel slots (c&1)*4..+3 of that row; A = weight row s*16 + (lane&15), chunk c (swizzled)
    const bool even_sw = HOT || ((p.stride_w | PWp) & 1) == 0;
    int xoff[WPX];
#pragma unroll
    for (int u = 0; u < WPX; u++)
        xoff[u] = ((wv * WPX + u) * p.stride_h + (lane >> 5)) * PWp + (lane & 15) * p.stride_w + ((lane >> 4) & 1) * 4; // dwords
    const int woff = lds_off(lane & 15, lane >> 4); // + s * 16 * BK for subtile s: 16 rows further the swizzle repeats
    unsigned t = blockIdx.x;
    int buf = 0;

    __syncthreads(); // zero fill of both patch buffers is complete
    if (t < ntiles) {
        fetch(t);
        commit(patch0);
    }
    ANAT_NOW(1);
    for (; t < ntiles; t += gridDim.x) {
        __syncthreads(); // patch[buf] committed by everyone; previous copy-out (tile, rowoff) finished
        const int8_t *patch = patch0 + buf * patch_bytes;
        const unsigned tn = t + gridDim.x;
        if (tn < ntiles) fetch(tn); // next tile's bytes travel while this one is computed

        // K loop.  The accumulators start as the C operand of the first step's MFMAs = the bias, read from LDS (a
        // global reload per tile would put a vmcnt(0) wait -- in-order counter -- between the next tile's fetch and
        // this tile's MFMAs; copying it into 32 accumulator registers first costs 32 moves per tile).  Operand
        // addresses: xoff[] (per lane, fixed for the whole run) + a scalar per (patch buffer, K step).
        v4i acc[WOC][WPX];
        auto kstep = [&](const int ks, const bool first) {
            const uint32_t *rows = (const uint32_t *)patch + ks * 2 * PWp;
            v4i xb[WPX];
#pragma unroll
            for (int u = 0; u < WPX; u++) {
                if (even_sw) { // 8-byte aligned: two ds_read_b64, conflict-free for 16 lanes at an 8-byte stride
                    const uint2 *q2 = (const uint2 *)(rows + xoff[u]);
                    const uint2 a0 = q2[0], a1 = q2[1];
                    xb[u] = (v4i){(int)a0.x, (int)a0.y, (int)a1.x, (int)a1.y};
                } else {
                    const uint32_t *q = rows + xoff[u];
                    xb[u] = (v4i){(int)q[0], (int)q[1], (int)q[2], (int)q[3]};
                } // row PH (odd-kh tail) exists and is zero
            }
#pragma unroll
            for (int s = 0; s < WOC; s++) {
                const v4i wa = *(const v4i *)(wl + ks * (BN * BK) + s * (16 * BK) + woff);
                if (first) {
                    const v4i b = sbias[s * 4 + (lane >> 4)];
#pragma unroll
                    for (int u = 0; u < WPX; u++) acc[s][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, xb[u], b, 0, 0, 0);
                } else {
#pragma unroll
                    for (int u = 0; u < WPX; u++) acc[s][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, xb[u], acc[s][u], 0, 0, 0);
                }
            }
        };
        kstep(0, true);
        for (int ks = 1; ks < nks; ks++) kstep(ks, false);
        if (tn < ntiles) commit(patch0 + (buf ^ 1) * patch_bytes); // last read before the previous epilogue's barrier
        buf ^= 1;
        int tx, ty;
        unsigned f;
        tile_xy(t, tx, ty, f);
        const int oy0 = ty * SC_TH, ox0 = tx * SC_TW, ow = p.out_w, oh = p.out_h;
        fill_rowoff<SC_BP>(p, rowoff,
                           [=](int row) {
                               const int oy = oy0 + (row >> 4), ox = ox0 + (row & 15);
                               return (oy < oh && ox < ow) ? (long)f * hw + (long)oy * ow + ox : -1L;
                           },
                           (unsigned)hw, dhw); // rewritten only after the next loop-top barrier
        __syncthreads();              // rowoff (and the committed next patch) visible to every wave
        if (HOT) { // host: NHWC rows stored straight from registers, half-step table, range fix-up dead, no fused Add
            __builtin_assume(p.lut2 != nullptr);
            __builtin_assume(p.add == nullptr);
            epilogue_t<SC_BP, BN, WPX, WOC, true, true, true, true>(p, acc, tile, slut, rowoff, 0, wv * (WPX * 16), 0, hw);
        } else {
            epilogue<SC_BP, BN, WPX, WOC, true>(p, acc, tile, slut, rowoff, 0, wv * (WPX * 16), 0, hw);
        }
    }
    ANAT_NOW(2);
    ANAT_END(p);
}

// ---------------------------------------------------------------------------------
// RGB stem, operand-direct form (in_c == 3, interleaved NHWC, stride 2 x even, kw <= 9: the hot case of the small-channel
// kernel).  In NHWC the kw*3 bytes a kernel row takes from the image are CONTIGUOUS: the MFMA B operand of output pixel
// (oy, ox), kernel row ky is just the 32 bytes at in[oy*2 - pt + ky][(ox*sw - pl)*3 ...] (bytes past kw*3 meet zero
// weights).  So nothing is staged: a wave owns 4 rows x 32 columns of a 16 x 32 output tile and loads its operands
// with buffer_load_dwordx4 -- per-lane offset fixed for the whole run, the tile in the scalar offset -- while it
// requantises the previous tile, takes the weights lane-linearly from LDS and stores every pixel's channels straight
// from registers.  No patch in LDS, no widening, no barrier, no row-offset table, and waves never wait for each other.
//  * One load serves several MFMAs: a K step covers kernel rows (2ks, 2ks+1), lanes 32-63 holding the odd row, and with
//    stride_h == 2 output row u reads image rows 2u + 2ks + {0,1}: the operand depends on u + ks only.  4 rows x 3
//    K steps need 6 row-pair loads per column class, not 12.
//  * Alignment decides the load rate (probed: a dwordx4 load runs at 64 B/clk when every lane's address is a multiple
//    of 4, at a quarter of that otherwise).  Pixels are 3 bytes, so with an even stride the 16 pixels of one MFMA are
//    the EVEN or the ODD columns of the tile (two column classes e): inside a class the byte address advances 6*sw per
//    pixel, a multiple of 4, and the class's residue d_e = (3*(sw*e - pl)) mod 4 is absorbed by loading from d_e bytes
//    earlier and using a copy of the weights shifted up by d_e bytes (18 + 3 <= 32: it fits the kernel row's K slot).
//    Tile origins advance 96*sw bytes and rows in_w*3: when in_w % 4 == 0 every load is aligned (otherwise still correct).
//  * Stores: a lane holds 8 channels of an even and of an odd column; v_permlane16_swap trades the halves between lane
//    rows g and g^1, so every lane stores 16 contiguous bytes and one instruction writes 1 KB of consecutive pixels
//    (8-byte stores of every other pixel doubled the L2 write requests: measured).
//  * The wave's 4 rows go in two phases (rows 0-1: MFMAs, requantise, store; rows 2-3 likewise) so that 32 accumulator
//    registers suffice and the next tile's loads are issued before the second phase's requantisation.
// Three fetch paths, chosen per wave and tile by scalar tests: INTERIOR (every tap inside the image); EDGE (rows outside
// the image get an out-of-range offset = zeros; bytes of columns outside the image are masked before the MFMAs -- they
// hold the neighbouring row's pixels); and the wave tiles whose 16-byte loads would start before / end after the tensor
// (two per batch) load from the nearest offset inside it and shift the bytes into place.
#define RGB_TW 32
#define RGB_TH 16
template <int WOC, int KS, bool LUT2>
__device__ __forceinline__ void conv_i8_rgb_body(
    const mhip_conv_i8_t &p, const int k64, const int tiles_x, const int tiles_y, const unsigned ntiles_all, const fastdiv_t dtx,
    const fastdiv_t dty, const unsigned in_bytes, const unsigned out_bytes) {
    ANAT_BEGIN();
    constexpr int TR = RGB_TH / 4;  // output rows per wave
    constexpr int NJ = TR + KS - 1; // row pairs of the wave's window
    extern __shared__ __attribute__((aligned(16))) int8_t dyn[];
    uint8_t *slut = (uint8_t *)dyn;          // the half-step table at LDS byte address 0 (requant_pack FAST)
    v4i *wl = (v4i *)(dyn + LUTB);           // [2 classes][KS][WOC][64 lanes]: A operands, lane-linear
    v4i *bl = wl + 2 * KS * WOC * 64;        // [WOC][4]: bias = C operand of the first K step
    lds_base_must_be_zero(dyn);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6); // scalar: tile offsets stay in SGPRs (a buffer's scalar offset
                                                             // computed from a vector value costs a waterfall loop per access)
    if (LUT2 && tid < 128) ((uint32_t *)slut)[tid] = ((const uint32_t *)p.lut2)[tid];
    const int dsh0 = (3 * (4 * p.stride_w - p.pad_left)) & 3, dsh1 = (3 * (5 * p.stride_w - p.pad_left)) & 3; // d_e (sw even)
    // weights: packed rows are [kh][8 taps x 4 bytes]; lane (m, g)'s A operand of (class e, K step ks, channel subtile s)
    // is bytes (g&1)*16 .. +15 of kernel row 2*ks + (g>>1), taps at 3 bytes each, moved up by d_e bytes
    if (p.w_rgb) { // laid out by the host at load time (mhip_conv_i8_rgb_pack): a plain copy
        for (int i = tid; i < 2 * KS * WOC * 64; i += NTHREADS) wl[i] = ((const v4i *)p.w_rgb)[i];
    } else {
        for (int i = tid; i < 2 * KS * WOC * 64 * 4; i += NTHREADS) {
            const int d = i & 3, l = (i >> 2) & 63, j = i >> 8, s2 = j % WOC, ks = (j / WOC) % KS, e = j / (WOC * KS);
            const int8_t *wrow = p.w + (size_t)(s2 * 16 + (l & 15)) * k64 + (2 * ks + (l >> 5)) * 32;
            uint32_t word = 0;
            for (int b = 0; b < 4; b++) {
                const int kb = ((l >> 4) & 1) * 16 + d * 4 + b - (e ? dsh1 : dsh0);
                if (kb >= 0 && kb < 3 * p.kw) word |= (uint32_t)(uint8_t)wrow[(kb / 3) * 4 + kb % 3] << (8 * b);
            }
            ((uint32_t *)wl)[i] = word;
        }
    }
    if (tid < WOC * 16) ((int *)bl)[tid] = p.bias ? p.bias[tid] : 0;

    const int n = lane & 15, g = lane >> 4, half = g & 1, kr = g >> 1;
    const int rowb = p.in_w * 3;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, (int)in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, 0, (int)out_bytes, 0x00020000);
    const int pixs = p.out_pix_stride ? p.out_pix_stride : p.out_c;
    // per-lane offset inside a wave tile's input window (never negative: a negative lane offset is out of range for the
    // buffer, whatever the scalar offset adds); what a column class adds goes into the scalar offset
    const int vin = kr * rowb + 2 * n * p.stride_w * 3 + half * 16;
    const int cls0 = -dsh0, cls1 = p.stride_w * 3 - dsh1;
    // per-lane offset inside an output row of the tile.  WOC == 2: after the lane swap this lane stores 16 channels of
    // column 2n + (g&1); WOC == 4: 16 channels of column 2n + e, once per class
    const int vout = (WOC == 2 ? (2 * n + (g & 1)) * pixs + (g >> 1) * 16 : 2 * n * pixs + g * 16) + p.out_ch_off;
    const int vch = WOC == 2 ? (g >> 1) * 16 : g * 16; // first channel this lane stores

    // tile order as in conv_i8_smallc: XCD x walks the x-th eighth of the tile list
    const unsigned xcd = blockIdx.x & 7u;
    const unsigned xstart = (unsigned)(((unsigned long long)ntiles_all * xcd) >> 3);
    const unsigned ntiles = ((unsigned)(((unsigned long long)ntiles_all * (xcd + 1u)) >> 3) - xstart) * 8u + xcd;
    auto tile_xy = [&](unsigned t, int &tx, int &ty, unsigned &f) {
        const unsigned j = xstart + (t >> 3), q = fdiv(j, dtx);
        tx = (int)(j - q * (unsigned)tiles_x);
        f = fdiv(q, dty);
        ty = (int)(q - f * (unsigned)tiles_y);
    };

    v4i xb[2][NJ];                                      // [class][row pair j]: image rows iy0 + 2j + kr
    bool masked = false;                                // wave-uniform: EDGE operands wait for their column mask
    // row pairs [J0, J1) of tile t's window
    auto fetch = [&](unsigned t, auto J0c, auto J1c) {
        constexpr int J0 = decltype(J0c)::value, J1 = decltype(J1c)::value;
        int tx, ty;
        unsigned f;
        tile_xy(t, tx, ty, f);
        const int iy0 = (ty * RGB_TH + wv * TR) * 2 - p.pad_top, ix0 = tx * RGB_TW * p.stride_w - p.pad_left;
        const long fbase = (long)f * (long)p.in_stride;
        const int iy_last = iy0 + 2 * NJ - 1; // last image row of the window
        const int iy_lastv = iy_last < p.in_h - 1 ? iy_last : p.in_h - 1;
        // last byte + 1 any lane with a row inside the image touches / first byte of the window's first such row
        const long reach_hi = fbase + (long)iy_lastv * rowb + (long)(ix0 + (RGB_TW - 1) * p.stride_w) * 3 + 32;
        const long reach_lo = fbase + (long)(iy0 > 0 ? iy0 : 0) * rowb + (long)ix0 * 3 - 3;
        const bool inside = reach_hi <= (long)in_bytes && reach_lo >= 0;
        const bool interior = iy0 >= 0 && iy_last < p.in_h && ix0 >= 0 && ix0 + (RGB_TW - 1) * p.stride_w + p.kw <= p.in_w;
        masked = !interior;
        if (interior && inside) { // scalar tile offset + fixed lane offset
            const unsigned sbase = (unsigned)(fbase + (long)iy0 * rowb + (long)ix0 * 3);
#pragma unroll
            for (int j = J0; j < J1; j++)
#pragma unroll
                for (int e = 0; e < 2; e++)
                    xb[e][j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, vin, (int)(sbase + (unsigned)(2 * j * rowb + (e ? cls1 : cls0))), 0);
            return;
        }
        if (inside) {
            const int sb = (int)(fbase + (long)iy0 * rowb + (long)ix0 * 3); // may be negative: goes into the lane offset
#pragma unroll
            for (int j = J0; j < J1; j++) {
                const bool rv = (unsigned)(iy0 + 2 * j + kr) < (unsigned)p.in_h;
                const int off = vin + sb + 2 * j * rowb;
#pragma unroll
                for (int e = 0; e < 2; e++)
                    xb[e][j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, rv ? off + (e ? cls1 : cls0) : -1, 0, 0);
            }
            return;
        }
        // a 16-byte load of this window would start before / end after the tensor (two wave tiles per batch): load from
        // the nearest offset that keeps all 16 bytes inside and shift the bytes into place (zeros move in; whatever lies
        // outside the lane's image row is masked like on every edge tile)
        masked = true;
        const long sb = fbase + (long)iy0 * rowb + (long)ix0 * 3;
#pragma unroll
        for (int j = J0; j < J1; j++) {
            const bool rv = (unsigned)(iy0 + 2 * j + kr) < (unsigned)p.in_h;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const long off = sb + vin + 2 * j * rowb + (e ? cls1 : cls0);
                long lo = off < 0 ? 0 : off;
                lo = lo > (long)in_bytes - 16 ? (long)in_bytes - 16 : lo;
                const int d = (int)(off - lo); // wanted byte b = loaded byte b + d
                const v4i v = __builtin_amdgcn_raw_buffer_load_b128(xrs, rv ? (int)lo : -1, 0, 0);
                unsigned __int128 w = ((unsigned __int128)(uint32_t)v[3] << 96) | ((unsigned __int128)(uint32_t)v[2] << 64) |
                                      ((unsigned __int128)(uint32_t)v[1] << 32) | (unsigned __int128)(uint32_t)v[0];
                if (d >= 16 || d <= -16) w = 0;
                else if (d > 0) w >>= 8 * d;
                else if (d < 0) w <<= -8 * d;
                xb[e][j] = (v4i){(int)(uint32_t)w, (int)(uint32_t)(w >> 32), (int)(uint32_t)(w >> 64), (int)(uint32_t)(w >> 96)};
            }
        }
    };

    // rows u0, u0 + 1 of the wave's tile: MFMAs over every K step, class and channel subtile.  The A operands come from LDS
    // one group ahead of their MFMAs and no further (left alone the scheduler hoists all twelve reads: 48 registers)
    auto rows_mfma = [&](int u0, v4i (&acc)[WOC][2][2]) {
        constexpr int NG = KS * WOC * 2;
        v4i wa = wl[lane]; // group 0 = (e 0, ks 0, s 0)
#pragma unroll
        for (int gi = 0; gi < NG; gi++) {
            const int ks = gi / (WOC * 2), s2 = (gi / 2) % WOC, e = gi & 1;
            v4i wn = wa;
            if (gi + 1 < NG) {
                const int ks1 = (gi + 1) / (WOC * 2), s1 = ((gi + 1) / 2) % WOC, e1 = (gi + 1) & 1;
                wn = wl[((e1 * KS + ks1) * WOC + s1) * 64 + lane];
            }
            if (ks == 0) {
                const v4i b4 = bl[s2 * 4 + g];
#pragma unroll
                for (int u = 0; u < 2; u++) acc[s2][u][e] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, xb[e][u0 + u + ks], b4, 0, 0, 0);
            } else {
#pragma unroll
                for (int u = 0; u < 2; u++)
                    acc[s2][u][e] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, xb[e][u0 + u + ks], acc[s2][u][e], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            wa = wn;
        }
    };
    // ... requantised, packed and stored
    auto rows_store = [&](int u0, v4i (&acc)[WOC][2][2], int oy0, int ox0, unsigned obase) {
        const bool chok = vch < p.out_c;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            uint32_t pk[2][WOC];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                int a[WOC * 4];
#pragma unroll
                for (int s2 = 0; s2 < WOC; s2++)
#pragma unroll
                    for (int r = 0; r < 4; r++) a[s2 * 4 + r] = acc[s2][u][e][r];
                if (LUT2) requant_pack<WOC * 4, true, true, true, false, true>(a, p.cs, -128, slut + 128, pk[e]);
                else requant_pack<WOC * 4, false, true, true>(a, p.cs, p.relu ? 0 : -128, slut + 128, pk[e]);
                __builtin_amdgcn_sched_barrier(0); // one class at a time: interleaved, the temporaries of all four cost a wave per SIMD
            }
            const int oy = oy0 + u0 + u;
            const int soff = (int)(obase + (unsigned)(oy * p.out_w) * (unsigned)pixs);
#ifdef RGB_ABL // timing-only build (tools/stamps_build.sh rgbabl 1): every store dropped by the buffer unit -- what the stem costs without its output
            const bool rok = false;
#else
            const bool rok = chok && oy < p.out_h; // stores always issue: the same vmcnt in every wave
#endif
            if (WOC == 2) {
                // lane rows g, g^1 trade halves: even g ends with channels 8g..8g+15 of column 2n, odd g with channels
                // 8(g-1)..8(g-1)+15 of column 2n+1
                const auto w0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                const auto w1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                const int voff = (rok && ox0 + 2 * n + (g & 1) < p.out_w) ? vout : -1;
                __builtin_amdgcn_raw_buffer_store_b128((v4i){(int)w0[0], (int)w1[0], (int)w0[1], (int)w1[1]}, ors, voff, soff, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int voff = (rok && ox0 + 2 * n + e < p.out_w) ? vout + e * pixs : -1;
                    __builtin_amdgcn_raw_buffer_store_b128((v4i){(int)pk[e][0], (int)pk[e][1], (int)pk[e][WOC > 2 ? 2 : 0], (int)pk[e][WOC > 3 ? 3 : 0]}, ors, voff, soff, 0);
                }
            }
        }
    };

    // The window's first NA row pairs (all that rows 0-1 need) are requested while the PREVIOUS tile's second phase
    // requantises, the rest after the tile's own first-phase MFMAs (they travel during its requantisation): at most 32
    // operand registers are live together with the 32 accumulators and the requantisation's temporaries.
    constexpr int NA = (1 + KS < NJ) ? 1 + KS : NJ;
    using jz = std::integral_constant<int, 0>;
    using ja = std::integral_constant<int, NA>;
    using jn = std::integral_constant<int, NJ>;
    auto mask_pairs = [&](int ox0, auto J0c, auto J1c) { // EDGE: columns outside the image delivered the neighbouring row's bytes
        constexpr int J0 = decltype(J0c)::value, J1 = decltype(J1c)::value;
        const int ix0 = ox0 * p.stride_w - p.pad_left;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            // this lane's 16 bytes start at byte b0 of its image row: bytes [nlo, nhi) are inside the row
            const int b0 = (ix0 + (2 * n + e) * p.stride_w) * 3 + half * 16 - (e ? dsh1 : dsh0);
            int nlo = -b0, nhi = rowb - b0;
            nlo = nlo < 0 ? 0 : (nlo > 16 ? 16 : nlo);
            nhi = nhi < 0 ? 0 : (nhi > 16 ? 16 : nhi);
            v4i keep;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const int a = nlo - 4 * d, b = nhi - 4 * d; // bytes [a, b) of dword d
                const uint32_t below_b = b >= 4 ? 0xFFFFFFFFu : (b <= 0 ? 0u : (1u << (8 * b)) - 1u);
                const uint32_t below_a = a >= 4 ? 0xFFFFFFFFu : (a <= 0 ? 0u : (1u << (8 * a)) - 1u);
                keep[d] = (int)(below_b & ~below_a);
            }
#pragma unroll
            for (int j = J0; j < J1; j++) xb[e][j] &= keep;
        }
    };

    __syncthreads(); // table, weights, bias in LDS
    unsigned t = blockIdx.x;
    if (t < ntiles) fetch(t, jz{}, ja{});
    ANAT_NOW(1);
    for (; t < ntiles; t += gridDim.x) {
        int tx, ty;
        unsigned f;
        tile_xy(t, tx, ty, f);
        const int oy0 = ty * RGB_TH + wv * TR, ox0 = tx * RGB_TW;
        const unsigned obase = f * (unsigned)p.out_stride + (unsigned)ox0 * (unsigned)pixs;
        const bool edge = masked;
        if (edge) mask_pairs(ox0, jz{}, ja{});
        v4i acc[WOC][2][2];
        rows_mfma(0, acc);
        if (NA < NJ) fetch(t, ja{}, jn{}); // (sets `masked` to the same value again)
        rows_store(0, acc, oy0, ox0, obase);
        if (edge && NA < NJ) mask_pairs(ox0, ja{}, jn{});
        rows_mfma(2, acc);
        const unsigned tn = t + gridDim.x;
        if (tn < ntiles) fetch(tn, jz{}, ja{}); // the next tile's operands travel during the second phase's requantisation
        rows_store(2, acc, oy0, ox0, obase);
    }
    ANAT_NOW(2);
    ANAT_END(p);
}
ANAT_SETTER(mhip_anatomy_set_stem)

// the hot instantiation (32 channels, fused table) fits 4 waves per SIMD without spilling; the others are left to the
// allocator (3 waves)
template <int WOC, int KS, bool LUT2>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv_i8_rgb4(
    const mhip_conv_i8_t p, const int k64, const int tiles_x, const int tiles_y, const unsigned ntiles_all, const fastdiv_t dtx,
    const fastdiv_t dty, const unsigned in_bytes, const unsigned out_bytes) {
    conv_i8_rgb_body<WOC, KS, LUT2>(p, k64, tiles_x, tiles_y, ntiles_all, dtx, dty, in_bytes, out_bytes);
}
template <int WOC, int KS, bool LUT2>
__global__ __launch_bounds__(NTHREADS) void conv_i8_rgb(const mhip_conv_i8_t p, const int k64, const int tiles_x, const int tiles_y,
                                                        const unsigned ntiles_all, const fastdiv_t dtx, const fastdiv_t dty,
                                                        const unsigned in_bytes, const unsigned out_bytes) {
    conv_i8_rgb_body<WOC, KS, LUT2>(p, k64, tiles_x, tiles_y, ntiles_all, dtx, dty, in_bytes, out_bytes);
}

// host twin of the loop above: the kernel's LDS weight image [2 classes][KS][WOC][64 lanes][16 bytes]
extern "C" size_t mhip_conv_i8_rgb_pack(int in_c, int kh, int kw, int stride_h, int stride_w, int pad_left, int oc_pad, int k64,
                                        const int8_t *packed, int8_t *out) {
    const int KS = (kh + 1) / 2, WOC = oc_pad / 16;
    if (in_c != 3 || kw > 9 || (stride_w & 1) || stride_h != 2 || KS < 1 || KS > 4 || k64 != KS * 64 || (WOC != 2 && WOC != 4)) return 0;
    const size_t bytes = (size_t)2 * KS * WOC * 64 * 16;
    if (!out) return bytes;
    const int dsh0 = (3 * (4 * stride_w - pad_left)) & 3, dsh1 = (3 * (5 * stride_w - pad_left)) & 3;
    for (size_t i = 0; i < bytes / 4; i++) {
        const int d = (int)(i & 3), l = (int)((i >> 2) & 63), j = (int)(i >> 8), s2 = j % WOC, ks = (j / WOC) % KS, e = j / (WOC * KS);
        const int8_t *wrow = packed + (size_t)(s2 * 16 + (l & 15)) * k64 + (2 * ks + (l >> 5)) * 32;
        uint32_t word = 0;
        for (int b = 0; b < 4; b++) {
            const int kb = ((l >> 4) & 1) * 16 + d * 4 + b - (e ? dsh1 : dsh0);
            if (kb >= 0 && kb < 3 * kw) word |= (uint32_t)(uint8_t)wrow[(kb / 3) * 4 + kb % 3] << (8 * b);
        }
        memcpy(out + i * 4, &word, 4);
    }
    return bytes;
}

static inline const tune_t &tune() { return conv_i8_tune_state(); }
// operand-direct RGB stem (conv_i8_rgb): -2 = not a shape it takes
template <int WOC>
static int try_rgb(const mhip_conv_i8_t *p, int k64) {
    const bool direct = !p->out_nchw && ((p->out_c | p->out_pix_stride | p->out_ch_off) & 15) == 0; // as epilogue()
    const bool rgb_ok = direct && p->in_c == 3 && p->safe && !p->add && (p->lut2 || !p->lut);
    // operand-direct form: 32-bit offsets, stride 2 x even, 16 stored channels per lane, the class shift inside the K slot
    const long in_ext = in_extent_bytes(p), out_ext = persist_out_bytes(p);
    const int ksteps = (p->kh + 1) / 2;
    if (rgb_ok && tune().rgb_direct && p->kw <= 9 && (p->stride_w & 1) == 0 && p->stride_h == 2 && p->out_c % 16 == 0 && ksteps >= 1 &&
        ksteps <= 4 && k64 == ksteps * 64 && in_ext >= 16 && in_ext < 0x7fffffffL && out_ext < 0x7fffffffL) {
        const int rtx = (p->out_w + RGB_TW - 1) / RGB_TW, rty = (p->out_h + RGB_TH - 1) / RGB_TH;
        const long rtiles = (long)rtx * rty * p->frames;
        if (rtiles >= 0x0fffffffL) return -1;
        const long rgrid = rtiles < 256L * 8 ? (rtiles + 7) / 8 * 8 : 256L * 8;
        const fastdiv_t dtx = make_fastdiv((unsigned)rtx), dty = make_fastdiv((unsigned)rty);
        const size_t rgb_lds = LUTB + 2 * (size_t)ksteps * WOC * 1024 + WOC * 64;
#define RGB(K)                                                                                                                   \
    hipLaunchKernelGGL((p->lut2 ? (WOC == 2 ? conv_i8_rgb4<WOC, K, true> : conv_i8_rgb<WOC, K, true>) : conv_i8_rgb<WOC, K, false>), \
                       dim3((unsigned)rgrid), dim3(NTHREADS), rgb_lds, mhip_stream_native(), *p, k64, rtx, rty, (unsigned)rtiles, dtx, \
                       dty, (unsigned)in_ext, (unsigned)out_ext)
        switch (ksteps) {
            case 1: RGB(1); break;
            case 2: RGB(2); break;
            case 3: RGB(3); break;
            default: RGB(4); break;
        }
#undef RGB
        return mhip_check(hipGetLastError(), "conv_i8_rgb launch");
    }
    return -2;
}

template <int WOC>
static int launch_smallc(const mhip_conv_i8_t *p, int k64) {
    const int tiles_x = (p->out_w + SC_TW - 1) / SC_TW, tiles_y = (p->out_h + SC_TH - 1) / SC_TH;
    const long ntiles = (long)tiles_x * tiles_y * p->frames;
    const int PH = (SC_TH - 1) * p->stride_h + p->kh, PW = (SC_TW - 1) * p->stride_w + p->kw;
    const int PWp = (PW + 8 + 3) & ~3;
    const int gpr = (PW + 3) / 4;
    if ((long)PH * gpr > 2 * NTHREADS || ntiles > 0x7fffffffL) return -1;
    constexpr int BN = WOC * 16;
    const bool direct = !p->out_nchw && ((p->out_c | p->out_pix_stride | p->out_ch_off) & 15) == 0; // as epilogue()
    const size_t tile_bytes = direct ? 0 : (size_t)SC_BP * (BN + OPAD);
    const size_t lds = (size_t)BN * k64 + 2 * ((((size_t)PH + 1) * PWp * 4 + 15) & ~(size_t)15) + tile_bytes + LUTB +
                       (size_t)SC_BP * 8 + (size_t)BN * 4;
    if (lds > 64 * 1024) return -1;
    if (ntiles >= 0x0fffffffL) return -1; // tile ids reach 8 x the longest per-XCD range
    long grid = ntiles < 256L * 8 ? (ntiles + 7) / 8 * 8 : 256L * 8; // a multiple of 8: a workgroup stays on its XCD's ids
    const bool hot = direct && p->in_c == 3 && p->in_w >= 4 && ((p->stride_w | PWp) & 1) == 0 && p->lut2 && p->safe && !p->add;
    auto kern = hot ? conv_i8_smallc<WOC, true> : conv_i8_smallc<WOC, false>;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTHREADS), lds, mhip_stream_native(), *p, k64,
                       tiles_x, tiles_y, (unsigned)ntiles, PH, PW, PWp, make_fastdiv((unsigned)(p->out_h * p->out_w)),
                       make_fastdiv((unsigned)tiles_x), make_fastdiv((unsigned)tiles_y), make_fastdiv((unsigned)gpr), (int)tile_bytes);
    return mhip_check(hipGetLastError(), "conv_i8_smallc launch");
}

int conv_i8_try_rgb(const mhip_conv_i8_t *p, int k64) { return p->oc_pad == 32 ? try_rgb<2>(p, k64) : try_rgb<4>(p, k64); }

// the patch-widening small-channel kernel: -2 when its patch / LDS budget does not take the shape (large strides or kernels)
int conv_i8_try_smallc(const mhip_conv_i8_t *p, int k64) {
    const int oc_pad = p->oc_pad;
    const int PH = (SC_TH - 1) * p->stride_h + p->kh, PW = (SC_TW - 1) * p->stride_w + p->kw;
    const int PWp = (PW + 8 + 3) & ~3;
    const bool direct = conv_i8_direct_rows(p);
    const size_t lds = (size_t)oc_pad * k64 + 2 * ((((size_t)PH + 1) * PWp * 4 + 15) & ~(size_t)15) +
                       (direct ? 0 : (size_t)SC_BP * (oc_pad + OPAD)) + LUTB + (size_t)SC_BP * 8 + (size_t)oc_pad * 4;
    const long ntiles = (long)((p->out_w + SC_TW - 1) / SC_TW) * ((p->out_h + SC_TH - 1) / SC_TH) * p->frames;
    if ((long)PH * ((PW + 3) / 4) > 2 * NTHREADS || lds > 64 * 1024 || ntiles >= 0x0fffffffL) return -2;
    // out_c <= 16 (the stem of the shipped yolov5n files): one channel subtile -- half the MFMAs and half the requantisation of the 32-row form
    static const int wide16 = getenv("MARS_HIP_SMALLC_WIDE16") != nullptr; // (A / B switch)
    if (p->out_c <= 16 && !wide16) return launch_smallc<1>(p, k64);
    return oc_pad == 32 ? launch_smallc<2>(p, k64) : launch_smallc<4>(p, k64);
}
