// conv_i8.hip -- int8 convolution for gfx950 (MI355X): im2col-free implicit
// GEMM on v_mfma_i32_16x16x64_i8.
//
// Replaces reference src/mars/mxu_conv.c:713-757 (conv2d_int8_nhwc_mxu) and,
// through the NCHW store mode, :630-670 (conv2d_int8_mxu).  Arithmetic
// contract:
//   acc  = bias[oc] + sum over in-image taps of in*w          (int32, exact)
//   r    = (int32)( (float)acc*cs + (scaled>=0 ? 0.5f : -0.5f) )   x86 truncation:
//          unrepresentable / NaN -> INT_MIN
//   out  = clamp(r, -128, 127)   [then max(.,0) if fused ReLU, then lut[.] if fused map]
// int32 accumulation is order independent, so the MFMA reduction order and the
// zero padding of K are exact.
//
// GEMM view:  D[oc][pixel] = sum_k W[oc][k] * X[pixel][k]
//   MFMA A operand = weights  (M = 16 output channels)
//   MFMA B operand = pixels   (N = 16 output pixels; all frames of the batch flattened)
//   K = kernel rows x (kw*in_c bytes padded to 16).  In NHWC the kw*in_c bytes
//   of one kernel row are CONTIGUOUS in the input, so a 16-byte K chunk of a
//   pixel is one 16-byte global access -- there is no im2col buffer anywhere.
//
// Kernels (all share the LDS-DMA staging, the source-side XOR swizzle that makes the ds_read_b128 fragment reads
// bank-conflict free, and the register epilogue: bias as the first MFMA's C operand, 6 VALU per value requantise,
// LUT gather with an immediate offset, packed 16-byte stores of consecutive channels):
//   conv_i8_mfma     in_c % 16 == 0, one BPX x BN tile per workgroup, 2-3 stage ring, counted vmcnt, one raw
//                    s_barrier per K step                                   -> deep K loops (3x3, >= 64 channels)
//   conv_i8_persist  the same inner loop walking a run of pixel tiles with cross-tile prefetch and buffer stores;
//                    SEG: input = never-materialised concat of up to 4 tensors -> 1x1 layers
//   conv_i8_patch    input patch of a tile staged once in LDS, weights resident, taps fed from LDS
//                                                                           -> k x k layers on wide maps
//   conv_i8_rgb      in_c == 3, stride 2 (the RGB stem): MFMA operands loaded straight from the image, no staging
//   conv_i8_smallc   in_c <= 4 otherwise: patch with pixels widened to 4 bytes, weights resident
//   conv_i8_generic  any other in_c: register-staged byte gather (fallback)
// Launch variants / policy / autotune hooks: bottom of this file.  Design notes: DESIGN.md section 5.
#include "conv_i8_common.hpp"

// ---------------------------------------------------------------------------------
// main kernel: in_c % 16 == 0.  BPX pixels x BN channels per workgroup; every wave owns a
// 64-pixel x (32|64)-channel accumulator tile (BPX=256 for BN<=64, BPX=128 for BN=128).
// POW2: in_c is a power of two (every yolov5 layer) -> the K position of a chunk is shifts and
// one small multiply instead of carried counters.
// KS = 64-byte K slices per ring stage: 2 halves the barriers and waits per MFMA (one s_barrier per 128 bytes of K)
// at twice the LDS per stage; the host picks it only for an even number of K steps.
// NW = waves per workgroup: 8 (512 threads) runs a 256 x 128 tile with the same 64 x 64 wave tiles, so the weight tile
// of a K step is shared by twice the pixels: 24 KB of LDS-DMA per 2.1 M MAC instead of 32 KB (the deep-K loop is bound
// by DMA latency x bytes in flight) at unchanged registers per wave.
#ifndef I8M_ABL // timing-only ablations of conv_i8_mfma (tools/stamps_build.sh i8m N; wrong bytes): 1 no epilogue (accumulators kept alive),
#define I8M_ABL 0 // 2 no K loop at all (prologue + epilogue), 4 no MFMAs, 8 no DMA
#endif
template <int BPX, int BN, int STAGES, int KS = 1, int NW = 4>
__global__ __launch_bounds__(NW * 64) void conv_i8_mfma(const mhip_conv_i8_t p, const long total_pix, const int k64,
                                                         const int8_t *__restrict__ zeros, const unsigned noc,
                                                         const unsigned nblk, const int lg_inc, const unsigned kw_magic,
                                                         const fastdiv_t dhw, const fastdiv_t dow, const int bufmode,
                                                         const unsigned in_bytes) {
    ANAT_BEGIN();
    const bool POW2 = lg_inc >= 0; // in_c is a power of two: K position by shifts, else carried counters
    const bool masked = p.kh * p.kw <= 32;
    constexpr int SLICE = (BPX + BN) * BK;
    constexpr int STAGE = KS * SLICE;
    constexpr int NWN = BN == 128 ? 2 : 1;       // waves along oc
    constexpr int NWM = NW / NWN;                // waves along pixels
    constexpr int WPX = BPX / NWM / 16;          // pixel subtiles per wave
    constexpr int WOC = BN / NWN / 16;           // oc subtiles per wave
    constexpr int XROWS = BPX / NW;              // X-tile rows a wave fetches
    constexpr int XI = XROWS / 16;               // X-tile DMA instructions per wave (16 rows each)
    constexpr int LW = (BN / 16 + NW - 1) / NW;  // W-tile DMA instructions per wave
    constexpr int L = XI + LW;                   // DMA instructions per wave per stage
    static_assert(XI >= 1 && XROWS % 16 == 0, "every wave fetches whole 16-row pieces of the pixel tile");
    // dynamic LDS: [lut 256 B][rowoff][ring: min(nks, STAGES) stages, reused as the output tile]
    extern __shared__ __attribute__((aligned(16))) int8_t dynlds[];
    uint8_t *slut = (uint8_t *)dynlds; // LDS byte address 0: this kernel owns no static LDS (requant_pack LUT0)
    lds_base_must_be_zero(dynlds);
    long *rowoff = (long *)(dynlds + LUTB);
    int8_t *lds = dynlds + BPX * 8 + LUTB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned id = xcd_remap(blockIdx.x, nblk);
    const long pix0 = (long)(id / noc) * BPX;
    const int oc0 = (int)(id % noc) * BN;
    const int hw = p.out_h * p.out_w;

    // bias first: the loads travel while the index math below runs; consumed by the first MFMA
    const int wm = wv % NWM, wn = wv / NWM;
    const int pxw = wm * (WPX * 16), ocw = wn * (WOC * 16);
    v4i acc[WOC][WPX];
    init_acc<WPX, WOC>(p, acc, oc0 + ocw);
    if (p.lut2) { if (tid < 128) ((uint32_t *)slut)[tid] = ((const uint32_t *)p.lut2)[tid]; }
    else if (p.lut && tid < 64) ((uint32_t *)slut)[tid] = ((const uint32_t *)p.lut)[tid];
    fill_rowoff<BPX>(p, rowoff, [=](int row) { long q = pix0 + row; return q < total_pix ? q : -1L; }, (unsigned)hw, dhw);

    // ---- DMA assignment.  One wave-instruction fills 16 consecutive 64-byte rows; lane i
    // lands in row i/4, slot i%4, so it must FETCH chunk (slot ^ swizzle(row)).
    const int schunk = (lane & 3) ^ (((lane >> 4) & 1) << 1);
    // Per row: pointer to the (possibly out-of-image) top-left input pixel of its window, and a bit mask
    // of the kernel taps that fall inside the image (bit ky*kw+kx), so the K loop spends one bit test per
    // row and step instead of four compares.  (kh*kw <= 32 is checked on the host; else MASKED is off.)
    const int8_t *xwin[XI];
    unsigned tapmask[XI];
    int iy0[XI], ix0[XI];
#pragma unroll
    for (int j = 0; j < XI; j++) {
        const long pix = pix0 + wv * XROWS + j * 16 + (lane >> 2);
        const bool valid = pix < total_pix;
        const unsigned f = valid ? fdiv((unsigned)pix, dhw) : 0u;
        const unsigned rem = valid ? (unsigned)pix - f * (unsigned)hw : 0u;
        const int oy = (int)fdiv(rem, dow), ox = (int)(rem - (unsigned)oy * (unsigned)p.out_w);
        iy0[j] = valid ? oy * p.stride_h - p.pad_top : -(1 << 28); // invalid rows fail every bounds test
        ix0[j] = ox * p.stride_w - p.pad_left;
        xwin[j] = p.in + (size_t)f * p.in_stride + ((long)iy0[j] * p.in_w + ix0[j]) * p.in_c;
        unsigned m = 0;
        if (masked) {
            // columns kx with 0 <= ix0+kx < in_w
            const int kx_lo = ix0[j] < 0 ? -ix0[j] : 0, kx_hi = p.in_w - ix0[j] < p.kw ? p.in_w - ix0[j] : p.kw; // [lo, hi)
            const unsigned colbits = kx_hi > kx_lo ? ((kx_hi >= 32 ? ~0u : (1u << kx_hi) - 1u) & ~((1u << kx_lo) - 1u)) : 0u;
            for (int r = 0; r < p.kh; r++) {
                const int iy = iy0[j] + r;
                if (iy >= 0 && iy < p.in_h) m |= colbits << (r * p.kw);
            }
        }
        tapmask[j] = m;
    }
    const int8_t *wsrc[LW];
    int wq[LW];
#pragma unroll
    for (int j = 0; j < LW; j++) {
        wq[j] = BN / 16 >= NW ? wv * LW + j : (wv % (BN / 16)); // fewer 16-row pieces than waves: some waves repeat one (same bytes, same place)
        wsrc[j] = p.w + (size_t)(oc0 + wq[j] * 16 + (lane >> 2)) * k64 + schunk * 16;
    }
    // BUF (host: in_c >= 64 and a power of two, tensors < 2 GiB, tap masks in use): a 64-byte K step lies inside ONE
    // tap, so tap / ky / kx / the step's byte offset are scalars, and the loads go through buffer resources with
    // 32-bit per-lane offsets -- ~8 instead of ~40 vector instructions per step (the K loop is issue bound)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, (int)in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, p.oc_pad * k64, 0x00020000);
    int xvoff[XI], wvoff[LW];
    if (bufmode) {
#pragma unroll
        for (int j = 0; j < XI; j++) xvoff[j] = (int)(xwin[j] - p.in) + schunk * 16; // may be negative: only used with in-image taps
#pragma unroll
        for (int j = 0; j < LW; j++) wvoff[j] = (oc0 + wq[j] * 16 + (lane >> 2)) * k64 + schunk * 16;
    }
    // K position of this lane's chunk: kernel row ky, tap kx, byte rc inside the tap
    int ky = 0, kx = 0, rc = schunk * 16;
    if (!POW2) {
        while (rc >= p.in_c) { rc -= p.in_c; kx++; }
        while (kx >= p.kw) { kx -= p.kw; ky++; }
    }
    const int taps = p.kh * p.kw;

    const int nks = k64 / BK;
    auto issue = [&](int ks, int stage) {
        int8_t *sb = lds + stage * STAGE + (KS > 1 ? (ks % KS) * SLICE : 0);
        if (bufmode) {
            const int utap = (ks * BK) >> lg_inc, urc = (ks * BK) & ((1 << lg_inc) - 1); // uniform
            const int uky = (int)(((unsigned)utap * kw_magic) >> 16), ukx = utap - uky * p.kw;
            const int ukoff = (uky * p.in_w + ukx) * p.in_c + urc;
            const bool uvalid = utap < taps;
#pragma unroll
            for (int j = 0; j < XI; j++) {
                const bool ok = uvalid & (((tapmask[j] >> utap) & 1u) != 0u);
                if (!(I8M_ABL & 8)) blds16(xrs, ok ? xvoff[j] + ukoff : -1, 0, sb + (wv * XROWS + j * 16) * BK);
            }
#pragma unroll
            for (int j = 0; j < LW; j++)
                if (!(I8M_ABL & 8)) blds16(wrs, wvoff[j], ks * BK, sb + BPX * BK + wq[j] * 16 * BK);
            return;
        }
        bool kvalid;
        if (POW2) {
            const unsigned pos = (unsigned)(ks * BK + schunk * 16);
            const unsigned tap = pos >> lg_inc;
            rc = (int)(pos & ((1u << lg_inc) - 1u));
            ky = (int)((tap * kw_magic) >> 16); // exact tap / kw for tap*(kw-1) < 65536 (host-checked)
            kx = (int)tap - ky * p.kw;
            kvalid = (int)tap < taps;
        } else {
            kvalid = ky < p.kh;
        }
        const long koff = ((long)ky * p.in_w + kx) * p.in_c + rc; // same for every row of this lane
        const int tap = ky * p.kw + kx;
#pragma unroll
        for (int j = 0; j < XI; j++) {
            bool ok;
            if (masked) {
                ok = kvalid & ((tapmask[j] >> tap) & 1u);
            } else {
                const int iy = iy0[j] + ky, ix = ix0[j] + kx;
                ok = kvalid & (iy >= 0) & (iy < p.in_h) & (ix >= 0) & (ix < p.in_w);
            }
            const int8_t *src = ok ? xwin[j] + koff : zeros;
            glds16(src, sb + (wv * XROWS + j * 16) * BK);
        }
#pragma unroll
        for (int j = 0; j < LW; j++) glds16(wsrc[j] + ks * BK, sb + BPX * BK + wq[j] * 16 * BK);
        if (!POW2) {
            rc += BK;
            while (rc >= p.in_c) { rc -= p.in_c; kx++; }
            while (kx >= p.kw) { kx -= p.kw; ky++; }
        }
    };

    const int nst = I8M_ABL & 2 ? 0 : nks / KS; // ring stages to run (KS == 2: the host guarantees an even nks)
#pragma unroll
    for (int s = 0; s < STAGES - 1; s++)
        if (s < nst)
#pragma unroll
            for (int u = 0; u < KS; u++) issue(s * KS + u, s);

    const int frow = lane & 15, fchunk = lane >> 4;
    int stage = 0, nstage = STAGES - 1;
    ANAT_NOW(1);
    for (int st = 0; st < nst; st++) {
        // stages still allowed in flight once stage st must have landed
        const int ahead = nst - 1 - st;
        if (STAGES >= 4 && ahead >= 2) wait_vmcnt<2 * L * KS>();
        else if (STAGES >= 3 && ahead >= 1) wait_vmcnt<L * KS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < KS; u++) {
            // all fragment reads of this slice first, then the next stage's DMA (its address math hides the LDS
            // latency), then the MFMAs
            const int8_t *xs = lds + stage * STAGE + u * SLICE, *ws = xs + BPX * BK;
            v4i xb[WPX], wa[WOC];
#pragma unroll
            for (int t = 0; t < WPX; t++) xb[t] = *(const v4i *)(xs + lds_off(pxw + t * 16 + frow, fchunk));
#pragma unroll
            for (int s = 0; s < WOC; s++) wa[s] = *(const v4i *)(ws + lds_off(ocw + s * 16 + frow, fchunk));
            if (u == 0 && st + STAGES - 1 < nst)
#pragma unroll
                for (int v = 0; v < KS; v++) issue((st + STAGES - 1) * KS + v, nstage);
#pragma unroll
            for (int s = 0; s < WOC; s++)
#pragma unroll
                for (int t = 0; t < WPX; t++) {
                    if (I8M_ABL & 4) asm volatile("" ::"v"(wa[s]), "v"(xb[t]));
                    else acc[s][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[s], xb[t], acc[s][t], 0, 0, 0);
                }
        }
        stage = stage + 1 == STAGES ? 0 : stage + 1;
        nstage = nstage + 1 == STAGES ? 0 : nstage + 1;
    }
    ANAT_NOW(2);
    __syncthreads(); // every wave is done reading the ring: reuse it for the output tile
    if (I8M_ABL & 1) {
        int fold = 0;
#pragma unroll
        for (int s = 0; s < WOC; s++)
#pragma unroll
            for (int t = 0; t < WPX; t++) fold ^= acc[s][t][0] ^ acc[s][t][1] ^ acc[s][t][2] ^ acc[s][t][3];
        if (fold == 0x12345678) p.out[0] = 1;
        return;
    }
    epilogue<BPX, BN, WPX, WOC, true>(p, acc, lds, slut, rowoff, oc0, pxw, ocw, hw);
    ANAT_END(p);
}

// ---------------------------------------------------------------------------------
// 128-byte K steps (in_c >= 128, a power of two): the deep-K kernels above are bound by L2 -> LDS DMA throughput, and
// that path delivers ~21-27 B/clk/CU when 4 lanes fetch a 64-byte piece of a row but ~40 when 8 lanes fetch a whole
// 128-byte line (tools/probes/probe_dmarate.hip).  Here a ring stage holds 128 bytes of K per row: one DMA instruction
// = 8 rows x 128 bytes (one request per cache line), two MFMA K-halves per stage, one barrier per 128 bytes of K.
// LDS rows are 128 bytes; 16-byte slot s of row r holds K chunk s ^ ((r >> 1) & 7) (source-side swizzle as before: the
// DMA writes lane-linearly), which gives the 16 lanes of every ds_read_b128 group 16 distinct bank groups:
// bank group = (r & 1) * 8 + slot, and rows of equal parity in a group carry 8 distinct values of chunk ^ (r >> 1).
// Tap decoding is scalar as in the buffer-addressed form (a 128-byte step lies inside one tap).
__device__ __forceinline__ int lds_off128(int row, int chunk8) { return row * 128 + ((chunk8 ^ ((row >> 1) & 7)) << 4); }
template <int BPX, int BN, int NW>
__global__ __launch_bounds__(NW * 64) void conv_i8_r128(const mhip_conv_i8_t p, const long total_pix, const int k128,
                                                         const unsigned noc, const unsigned nblk, const int lg_inc,
                                                         const unsigned kw_magic, const fastdiv_t dhw, const fastdiv_t dow,
                                                         const unsigned in_bytes) {
    constexpr int STAGE = (BPX + BN) * 128;
    constexpr int NWN = BN == 128 ? 2 : 1;
    constexpr int NWM = NW / NWN;
    constexpr int WPX = BPX / NWM / 16;
    constexpr int WOC = BN / NWN / 16;
    constexpr int XI = BPX / 8 / NW;  // X-tile DMA instructions per wave and stage (8 rows each)
    constexpr int WI = BN / 8 / NW;   // W-tile ...
    static_assert(XI >= 1 && WI >= 1 && BPX % (8 * NW) == 0 && BN % (8 * NW) == 0, "whole 8-row pieces per wave");
    extern __shared__ __attribute__((aligned(16))) int8_t dynlds[];
    uint8_t *slut = (uint8_t *)dynlds; // LDS byte address 0 (requant_pack LUT0)
    lds_base_must_be_zero(dynlds);
    long *rowoff = (long *)(dynlds + LUTB);
    int8_t *lds = dynlds + BPX * 8 + LUTB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned id = xcd_remap(blockIdx.x, nblk);
    const long pix0 = (long)(id / noc) * BPX;
    const int oc0 = (int)(id % noc) * BN;
    const int hw = p.out_h * p.out_w;
    const int wm = wv % NWM, wn = wv / NWM;
    const int pxw = wm * (WPX * 16), ocw = wn * (WOC * 16);
    v4i acc[WOC][WPX];
    init_acc<WPX, WOC>(p, acc, oc0 + ocw);
    if (p.lut2) { if (tid < 128) ((uint32_t *)slut)[tid] = ((const uint32_t *)p.lut2)[tid]; }
    else if (p.lut && tid < 64) ((uint32_t *)slut)[tid] = ((const uint32_t *)p.lut)[tid];
    fill_rowoff<BPX>(p, rowoff, [=](int row) { long q = pix0 + row; return q < total_pix ? q : -1L; }, (unsigned)hw, dhw);

    // DMA assignment: instruction j of this wave covers tile rows (wv * XI + j) * 8 .. + 7; lane i -> row i / 8, slot i % 8
    const int lrow = lane >> 3, lslot = lane & 7;
    int xvoff[XI];
    unsigned tapmask[XI];
#pragma unroll
    for (int j = 0; j < XI; j++) {
        const int trow = (wv * XI + j) * 8 + lrow;
        const long pix = pix0 + trow;
        const bool valid = pix < total_pix;
        const unsigned f = valid ? fdiv((unsigned)pix, dhw) : 0u;
        const unsigned rem = valid ? (unsigned)pix - f * (unsigned)hw : 0u;
        const int oy = (int)fdiv(rem, dow), ox = (int)(rem - (unsigned)oy * (unsigned)p.out_w);
        const int iy0 = oy * p.stride_h - p.pad_top, ix0 = ox * p.stride_w - p.pad_left;
        const int chunk = lslot ^ ((trow >> 1) & 7);
        xvoff[j] = (int)(f * (unsigned)p.in_stride) + (iy0 * p.in_w + ix0) * p.in_c + chunk * 16; // may be negative: only used with in-image taps
        const int kx_lo = ix0 < 0 ? -ix0 : 0, kx_hi = p.in_w - ix0 < p.kw ? p.in_w - ix0 : p.kw;
        const unsigned colbits = kx_hi > kx_lo ? ((kx_hi >= 32 ? ~0u : (1u << kx_hi) - 1u) & ~((1u << kx_lo) - 1u)) : 0u;
        unsigned m = 0;
        for (int r = 0; r < p.kh; r++) {
            const int iy = iy0 + r;
            if (iy >= 0 && iy < p.in_h) m |= colbits << (r * p.kw);
        }
        tapmask[j] = valid ? m : 0u;
    }
    int wvoff[WI];
#pragma unroll
    for (int j = 0; j < WI; j++) {
        const int trow = (wv * WI + j) * 8 + lrow;
        wvoff[j] = (oc0 + trow) * k128 + (lslot ^ ((trow >> 1) & 7)) * 16;
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, (int)in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, p.oc_pad * k128, 0x00020000);
    const int taps = p.kh * p.kw;
    const int nst = k128 / 128;
    auto issue = [&](int st, int buf) {
        int8_t *sb = lds + buf * STAGE;
        const int utap = (st * 128) >> lg_inc, urc = (st * 128) & ((1 << lg_inc) - 1); // uniform: a 128-byte step lies inside one tap
        const int uky = (int)(((unsigned)utap * kw_magic) >> 16), ukx = utap - uky * p.kw;
        const int ukoff = (uky * p.in_w + ukx) * p.in_c + urc;
        const bool uvalid = utap < taps;
#pragma unroll
        for (int j = 0; j < XI; j++) {
            const bool ok = uvalid & (((tapmask[j] >> utap) & 1u) != 0u);
            blds16(xrs, ok ? xvoff[j] + ukoff : -1, 0, sb + (wv * XI + j) * 1024);
        }
#pragma unroll
        for (int j = 0; j < WI; j++) blds16(wrs, wvoff[j], st * 128, sb + BPX * 128 + (wv * WI + j) * 1024);
    };

    issue(0, 0);
    const int frow = lane & 15, fchunk = lane >> 4;
    for (int st = 0; st < nst; st++) {
        wait_vmcnt<0>(); // stage st has landed (nothing else is in flight: the next stage is issued below)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int8_t *xs = lds + (st & 1) * STAGE, *ws = xs + BPX * 128;
        v4i xb[2][WPX], wa[2][WOC];
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int t = 0; t < WPX; t++) xb[h][t] = *(const v4i *)(xs + lds_off128(pxw + t * 16 + frow, h * 4 + fchunk));
#pragma unroll
            for (int q = 0; q < WOC; q++) wa[h][q] = *(const v4i *)(ws + lds_off128(ocw + q * 16 + frow, h * 4 + fchunk));
        }
        if (st + 1 < nst) issue(st + 1, (st + 1) & 1); // the other buffer: every wave is past its reads of stage st - 1
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int q = 0; q < WOC; q++)
#pragma unroll
                for (int t = 0; t < WPX; t++) acc[q][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[h][q], xb[h][t], acc[q][t], 0, 0, 0);
    }
    __syncthreads(); // every wave is done reading the ring: reuse it for the output tile
    epilogue<BPX, BN, WPX, WOC, true>(p, acc, lds, slut, rowoff, oc0, pxw, ocw, hw);
}

// ---------------------------------------------------------------------------------
// persistent form of the main kernel (the default for NHWC outputs with 16-byte aligned rows):
// a workgroup owns ONE channel tile and walks a contiguous run of pixel tiles.  The K stages of
// all its tiles form one stream through the LDS ring, so the loads of tile i+1 are in flight
// while tile i is still in its MFMAs and its epilogue; bias, LUT and weight pointers are set up
// once.  Results leave as raw buffer stores straight from registers (lanes outside the image get
// an out-of-range offset, which the buffer unit drops) -- always WPX stores per wave and tile, so
// the number of vector-memory operations younger than a given stage is known and vmcnt can be
// counted across the stores.  (gfx9: loads and stores retire in order against one vmcnt.)
template <int A, int B>
__device__ __forceinline__ void wait_vmcnt_at_most(int n) { // largest known-safe immediate <= n
    constexpr int HI = A > B ? A : B, LO = A > B ? B : A;
    if (n >= A + B) wait_vmcnt<A + B>();
    else if (n >= HI) wait_vmcnt<HI>();
    else if (n >= LO) wait_vmcnt<LO>();
    else wait_vmcnt<0>();
}

// SEG: the input is the channel concatenation of up to 4 tensors that was never materialised (1x1 convolutions only):
// K step ks reads its 64 channels from the tensor(s) that own them.  Segment boundaries are multiples of 32 channels,
// so the two 32-byte halves of a K step each lie in one segment and the choice is a scalar select plus one per-lane
// select between the halves.
// PAIR: two convolutions over the SAME input and geometry (C3's cv1 and cv2) in one launch: channel tiles
// [0, noc0) belong to the first, the rest to the second (`alt` = its output side).  The two workgroups that need a
// pixel tile are neighbours in the XCD-aware block order, so the second one's input reads hit L2: the input is
// read from HBM once instead of twice.
struct conv_out_side_t {
    int8_t *out;
    size_t out_stride;
    const int8_t *w;
    const int32_t *bias;
    const uint8_t *lut, *lut2;
    int out_c, relu, out_pix_stride, out_ch_off;
    float cs;
    unsigned out_bytes;
};
template <int BPX, int BN, int STAGES, bool HAS_LUT, bool SEG = false, bool PAIR = false>
__global__ __launch_bounds__(NTHREADS) void conv_i8_persist(const mhip_conv_i8_t pin, const unsigned total_pix, const int k64,
                                                            const int8_t *__restrict__ zeros, const unsigned noc,
                                                            const unsigned npt, const unsigned ngrp, const int lg_inc,
                                                            const unsigned kw_magic, const fastdiv_t dhw, const fastdiv_t dow,
                                                            const unsigned out_bytes_first, const conv_out_side_t alt,
                                                            const unsigned noc0, const int bufmode, const unsigned in_bytes,
                                                            const int wres) {
    // wres (host: 2-stage ring only): the weights of this workgroup's channel tile, all K steps, are fetched ONCE
    // into LDS behind the ring and stay there while the workgroup walks its pixel tiles; the ring then carries pixel
    // tiles only -- for 1x1 layers that halves the LDS-DMA bytes per tile
    ANAT_BEGIN();
    const int STAGE = wres ? BPX * BK : (BPX + BN) * BK;
    constexpr int NWN = BN == 128 ? 2 : 1;
    constexpr int NWM = 4 / NWN;
    constexpr int WPX = BPX / NWM / 16;
    constexpr int WOC = BN / NWN / 16;
    constexpr int XI = BPX / 64;
    constexpr int LW = BN >= 128 ? 2 : 1;
    constexpr int L = XI + LW;  // vector-memory instructions per wave per stage
    constexpr int NST = WPX;    // ... and per tile epilogue
    extern __shared__ __attribute__((aligned(16))) int8_t dynlds[];
    uint8_t *slut = (uint8_t *)dynlds; // LDS byte address 0 (requant_pack LUT0)
    int8_t *lds = dynlds + LUTB;
    lds_base_must_be_zero(dynlds);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned id = xcd_remap(blockIdx.x, noc * ngrp);
    const unsigned grp = id / noc;
    unsigned ot = id - grp * noc;
    mhip_conv_i8_t p = pin; // uniform; the second convolution of a pair swaps in its output side
    unsigned out_bytes = out_bytes_first;
    if (PAIR && ot >= noc0) {
        ot -= noc0;
        p.lut2 = alt.lut2; // its own half-step table (the two convolutions have different scales)
        p.out = alt.out; p.out_stride = alt.out_stride; p.w = alt.w; p.bias = alt.bias; p.lut = alt.lut;
        p.out_c = alt.out_c; p.relu = alt.relu; p.out_pix_stride = alt.out_pix_stride; p.out_ch_off = alt.out_ch_off;
        p.cs = alt.cs;
        out_bytes = alt.out_bytes;
    }
    const int oc0 = (int)ot * BN;
    const unsigned t0 = (unsigned)(((unsigned long long)grp * npt) / ngrp);
    const unsigned t1 = (unsigned)(((unsigned long long)(grp + 1) * npt) / ngrp);
    if (t0 >= t1) return;
    const unsigned hw = (unsigned)(p.out_h * p.out_w);
    const int wm = wv % NWM, wn = wv / NWM;
    const int pxw = wm * (WPX * 16), ocw = wn * (WOC * 16);

    // BN == 16 (round 6: out_c <= 16 -- the yolov5n models' first C3 -- ran 32-row tiles, half of every MFMA and of the requantisation for
    // channels that do not exist): ONE channel subtile.  The host packed weight rows and bias for two subtiles (MFMA row 4 g + r of subtile
    // s carries channel 8 g + 4 s + r: conv_i8_common.hpp); this form wants channel c in row c and reads packed row 16 s + 4 g + r for it.
    auto packed_row16 = [](int c) { return ((c >> 2) & 1) * 16 + (c >> 3) * 4 + (c & 3); };
    v4i bias[WOC];
#pragma unroll
    for (int s = 0; s < WOC; s++)
        bias[s] = p.bias ? *(const v4i *)(p.bias + (BN == 16 ? packed_row16((lane >> 4) * 4) : oc0 + ocw + s * 16 + (lane >> 4) * 4)) : (v4i){0, 0, 0, 0};
    if (HAS_LUT) {
        if (p.lut2) { if (tid < 128) ((uint32_t *)slut)[tid] = ((const uint32_t *)p.lut2)[tid]; }
        else if (tid < 64) ((uint32_t *)slut)[tid] = ((const uint32_t *)p.lut)[tid];
        __syncthreads();
    }
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)out_bytes, 0x00020000);

    const int schunk = (lane & 3) ^ (((lane >> 4) & 1) << 1);
    const int8_t *xwin[XI];
    unsigned tapmask[XI];
    unsigned rowf[SEG ? XI : 1], rowrem[SEG ? XI : 1]; // SEG: frame and pixel-in-frame of the rows this lane fetches
    unsigned rowrem_up[SEG ? XI : 1];                  // ... and the pixel a 2x nearest-upsampled segment reads instead
    int xvoff[XI];
    auto setup_rows = [&](unsigned tile) { // window origin and in-image tap mask of the rows this lane fetches
#pragma unroll
        for (int j = 0; j < XI; j++) {
            const unsigned pix = tile * BPX + wv * (BPX / 4) + j * 16 + (lane >> 2);
            const bool valid = pix < total_pix;
            const unsigned f = valid ? fdiv(pix, dhw) : 0u;
            const unsigned rem = valid ? pix - f * hw : 0u;
            if (SEG) {
                rowf[j] = f;
                rowrem[j] = rem;
                const unsigned oy = fdiv(rem, dow), ox = rem - oy * (unsigned)p.out_w;
                rowrem_up[j] = (oy >> 1) * ((unsigned)p.out_w >> 1) + (ox >> 1);
                tapmask[j] = valid ? 1u : 0u;
                continue;
            }
            const int oy = (int)fdiv(rem, dow), ox = (int)(rem - (unsigned)oy * (unsigned)p.out_w);
            const int iy0 = oy * p.stride_h - p.pad_top, ix0 = ox * p.stride_w - p.pad_left;
            xwin[j] = p.in + (size_t)f * p.in_stride + ((long)iy0 * p.in_w + ix0) * p.in_c;
            xvoff[j] = (int)(f * (unsigned)p.in_stride) + (iy0 * p.in_w + ix0) * p.in_c + schunk * 16; // BUF mode (32-bit offsets)
            const int kx_lo = ix0 < 0 ? -ix0 : 0, kx_hi = p.in_w - ix0 < p.kw ? p.in_w - ix0 : p.kw;
            const unsigned colbits = kx_hi > kx_lo ? ((kx_hi >= 32 ? ~0u : (1u << kx_hi) - 1u) & ~((1u << kx_lo) - 1u)) : 0u;
            unsigned m = 0;
            for (int r = 0; r < p.kh; r++) {
                const int iy = iy0 + r;
                if (iy >= 0 && iy < p.in_h) m |= colbits << (r * p.kw);
            }
            tapmask[j] = valid ? m : 0u;
        }
    };
    const int8_t *wsrc[LW];
    int wq[LW], wvoff[LW];
#pragma unroll
    for (int j = 0; j < LW; j++) {
        wq[j] = BN >= 64 ? wv * LW + j : (BN == 16 ? 0 : (wv & 1));
        const int wrow = BN == 16 ? packed_row16(lane >> 2) : oc0 + wq[j] * 16 + (lane >> 2);
        wsrc[j] = p.w + (size_t)wrow * k64 + schunk * 16;
        wvoff[j] = wrow * k64 + schunk * 16;
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, (int)in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, p.oc_pad * k64, 0x00020000);
    const int taps = p.kh * p.kw;
    const int nks = k64 / BK;

    auto issue = [&](int ks, int stage) {
        int8_t *sb = lds + stage * STAGE;
        const unsigned pos = (unsigned)(ks * BK + schunk * 16);
        if (SEG) {
            // owner of channel c: the last segment that starts at or before it (unused segments start at INT_MAX)
            auto owner = [&](int c) { return (c >= p.seg_c0[1]) + (c >= p.seg_c0[2]) + (c >= p.seg_c0[3]); };
            const int sa = owner(ks * BK), sb2 = owner(ks * BK + 32); // uniform
            const bool hi = schunk >= 2;
            const int8_t *base = hi ? p.seg_in[sb2] : p.seg_in[sa];
            const unsigned fstride = (unsigned)(hi ? p.seg_stride[sb2] : p.seg_stride[sa]);
            const unsigned segc = (unsigned)(hi ? p.seg_c[sb2] : p.seg_c[sa]);
            const unsigned coff = pos - (unsigned)(hi ? p.seg_c0[sb2] : p.seg_c0[sa]);
            const bool up = ((p.seg_up >> (hi ? sb2 : sa)) & 1) != 0; // segment = 2x nearest upsample of its tensor
            const bool kvalid = (int)pos < p.in_c;
#pragma unroll
            for (int j = 0; j < XI; j++) {
                const bool ok = kvalid & (tapmask[j] != 0u);
                const int8_t *src = base + (size_t)rowf[j] * fstride + (size_t)(up ? rowrem_up[j] : rowrem[j]) * segc + coff;
                glds16(ok ? src : zeros, sb + (wv * (BPX / 4) + j * 16) * BK);
            }
        } else if (bufmode) { // see conv_i8_mfma: scalar tap, buffer-addressed loads, zero fill by range check
            const int utap = (ks * BK) >> lg_inc, urc = (ks * BK) & ((1 << lg_inc) - 1);
            const int uky = (int)(((unsigned)utap * kw_magic) >> 16), ukx = utap - uky * p.kw;
            const int ukoff = (uky * p.in_w + ukx) * p.in_c + urc;
            const bool uvalid = utap < taps;
#pragma unroll
            for (int j = 0; j < XI; j++) {
                const bool ok = uvalid & (((tapmask[j] >> utap) & 1u) != 0u);
                blds16(xrs, ok ? xvoff[j] + ukoff : -1, 0, sb + (wv * (BPX / 4) + j * 16) * BK);
            }
            if (!wres)
#pragma unroll
                for (int j = 0; j < LW; j++) blds16(wrs, wvoff[j], ks * BK, sb + BPX * BK + wq[j] * 16 * BK);
            return;
        } else {
            const unsigned tap = pos >> lg_inc;
            const int rc = (int)(pos & ((1u << lg_inc) - 1u));
            const int ky = (int)((tap * kw_magic) >> 16);
            const int kx = (int)tap - ky * p.kw;
            const bool kvalid = (int)tap < taps;
            const long koff = ((long)ky * p.in_w + kx) * p.in_c + rc;
#pragma unroll
            for (int j = 0; j < XI; j++) {
                const bool ok = kvalid & ((tapmask[j] >> tap) & 1u);
                glds16(ok ? xwin[j] + koff : zeros, sb + (wv * (BPX / 4) + j * 16) * BK);
            }
        }
        if (!wres)
#pragma unroll
            for (int j = 0; j < LW; j++) glds16(wsrc[j] + ks * BK, sb + BPX * BK + wq[j] * 16 * BK);
    };

    // issue cursor over the (tile, k-step) stream; younger[i] = vector-memory instructions this wave has
    // issued after the i-th oldest stage still in the ring (a stage that was not issued counts as empty)
    int8_t *const wres_base = lds + STAGES * STAGE;
    if (wres)
        for (int ks = 0; ks < nks; ks++)
#pragma unroll
            for (int j = 0; j < LW; j++) glds16(wsrc[j] + ks * BK, wres_base + (ks * BN + wq[j] * 16) * BK);
    unsigned itile = t0;
    int iks = 0;
    int younger[STAGES - 1];
    setup_rows(t0);
    int nstage = 0;
#pragma unroll
    for (int s = 0; s < STAGES - 1; s++) {
        int n = 0;
        if (itile < t1) {
            issue(iks, nstage);
            n = wres ? XI : L;
            if (++iks == nks) {
                iks = 0;
                if (++itile < t1) setup_rows(itile);
            }
        }
#pragma unroll
        for (int i = 0; i < s; i++) younger[i] += n;
        younger[s] = 0;
        nstage++;
    }
    nstage = STAGES - 1;

    const int frow = lane & 15, fchunk = lane >> 4;
    const int chan = ocw + (lane >> 4) * (4 * WOC);
    const int pstride = p.out_pix_stride ? p.out_pix_stride : p.out_c;
    const int lo = p.relu ? 0 : -128;
    const uint8_t *lut128 = slut + 128;
    const bool ragged = (p.out_c % (WOC * 4)) != 0;
    int stage = 0;
    ANAT_NOW(1);
    for (unsigned tile = t0; tile < t1; tile++) {
        v4i acc[WOC][WPX];
        for (int ks = 0; ks < nks; ks++) {
            wait_vmcnt_at_most<(STAGES - 2) * L, NST>(younger[0]);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // all fragment reads of this step first, then the next stage's DMA (its address math hides the LDS
            // latency), then the MFMAs
            const int8_t *xs = lds + stage * STAGE, *ws = wres ? wres_base + ks * (BN * BK) : xs + BPX * BK;
            v4i xb[WPX], wa[WOC];
#pragma unroll
            for (int t = 0; t < WPX; t++) xb[t] = *(const v4i *)(xs + lds_off(pxw + t * 16 + frow, fchunk));
#pragma unroll
            for (int s = 0; s < WOC; s++) wa[s] = *(const v4i *)(ws + lds_off(ocw + s * 16 + frow, fchunk));
            int n = 0;
            if (itile < t1) {
                issue(iks, nstage);
                n = wres ? XI : L;
                if (++iks == nks) {
                    iks = 0;
                    if (++itile < t1) setup_rows(itile);
                }
            }
#pragma unroll
            for (int i = 0; i + 1 < STAGES - 1; i++) younger[i] = younger[i + 1] + n;
            younger[STAGES - 2] = 0;
            if (ks == 0) { // the first step of a tile takes the bias as its C operand: accumulators start there
#pragma unroll
                for (int s = 0; s < WOC; s++)
#pragma unroll
                    for (int t = 0; t < WPX; t++) acc[s][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[s], xb[t], bias[s], 0, 0, 0);
            } else {
#pragma unroll
                for (int s = 0; s < WOC; s++)
#pragma unroll
                    for (int t = 0; t < WPX; t++) acc[s][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[s], xb[t], acc[s][t], 0, 0, 0);
            }
            stage = stage + 1 == STAGES ? 0 : stage + 1;
            nstage = nstage + 1 == STAGES ? 0 : nstage + 1;
        }
        // epilogue: requantise, optional LUT, one buffer store per pixel subtile
#ifdef ANATOMY
        if (tile + 1 == t1) ANAT_NOW(2); // (the last tile's epilogue has no next tile's K steps to hide behind)
#endif
#pragma unroll
        for (int t = 0; t < WPX; t++) {
            const unsigned pix = tile * BPX + pxw + t * 16 + (lane & 15);
            const unsigned f = fdiv(pix, dhw), rem = pix - f * hw;
            const unsigned off = f * (unsigned)p.out_stride + rem * (unsigned)pstride + (unsigned)(p.out_ch_off + oc0 + chan);
            const int nvalid = p.out_c - (oc0 + chan); // channels of this lane's run that exist
            const bool ok = pix < total_pix && nvalid >= WOC * 4;
            uint32_t pk[WOC];
            int a[WOC * 4];
#pragma unroll
            for (int s = 0; s < WOC; s++)
#pragma unroll
                for (int r = 0; r < 4; r++) a[s * 4 + r] = acc[s][t][r];
            if (HAS_LUT && p.lut2) requant_pack<WOC * 4, HAS_LUT, true, true, false, true>(a, p.cs, lo, lut128, pk);
            else requant_pack<WOC * 4, HAS_LUT, true, true>(a, p.cs, lo, lut128, pk);
            const int voff = ok ? (int)off : -1; // 0xffffffff >= num_records: dropped by the buffer unit
            if (WOC == 4)
                __builtin_amdgcn_raw_buffer_store_b128((v4i){(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]}, orsrc, voff, 0, 0);
            else if (WOC == 2)
                __builtin_amdgcn_raw_buffer_store_b64((v2i){(int)pk[0], (int)pk[WOC > 1 ? 1 : 0]}, orsrc, voff, 0, 0);
            else
                __builtin_amdgcn_raw_buffer_store_b32((int)pk[0], orsrc, voff, 0, 0);
            if (ragged) {
                // out_c is not a multiple of the run (the 255-channel heads): the lane that holds the last, partial
                // run writes it as 8 + 4 + 2 + 1 bytes; every other lane's pieces go out of range.  Always the same
                // number of store instructions, so the vmcnt bookkeeping stays exact.
                const bool part = pix < total_pix && nvalid > 0 && nvalid < WOC * 4;
                const int nv = part ? nvalid : 0;
                const int o4 = nv & 8, o2 = nv & 12, o1 = nv & 14;
                const uint32_t w4 = o4 ? pk[(WOC * 4 > 8) ? 2 : 0] : pk[0];
                const int i2 = o2 >> 2, i1 = o1 >> 2;
                uint32_t w2 = pk[0], w1 = pk[0];
#pragma unroll
                for (int q = 1; q < WOC; q++) {
                    w2 = i2 == q ? pk[q] : w2;
                    w1 = i1 == q ? pk[q] : w1;
                }
                w1 >>= (o1 & 2) * 8;
                if (WOC == 4)
                    __builtin_amdgcn_raw_buffer_store_b64((v2i){(int)pk[0], (int)pk[1]}, orsrc, (nv & 8) ? (int)off : -1, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32((int)w4, orsrc, (nv & 4) ? (int)off + o4 : -1, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b16((short)w2, orsrc, (nv & 2) ? (int)off + o2 : -1, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b8((char)w1, orsrc, (nv & 1) ? (int)off + o1 : -1, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < STAGES - 1; i++) younger[i] += ragged ? NST * (WOC == 4 ? 5 : 4) : NST;
    }
    ANAT_END(p);
}
ANAT_SETTER(mhip_anatomy_set_i8)
// ---------------------------------------------------------------------------------
// generic kernel: any in_c (the 3-channel stem); register-staged byte gather
__device__ __forceinline__ bool mhip_small_c_dev(int in_c, int kw, int out_c) { return in_c <= 4 && kw <= 8 && out_c <= 64; }
template <int BN>
__global__ __launch_bounds__(NTHREADS) void conv_i8_generic(const mhip_conv_i8_t p, const long total_pix, const int k64,
                                                            const fastdiv_t dhw) {
    constexpr int STAGE = (BP + BN) * BK;
    constexpr int WPX = 2, WOC = BN / 16;
    constexpr int TILE_BYTES = BP * (BN + OPAD);
    constexpr int LDS_BYTES = (2 * STAGE > TILE_BYTES ? 2 * STAGE : TILE_BYTES) + 256;
    __shared__ __attribute__((aligned(16))) int8_t lds[LDS_BYTES];
    __shared__ long rowoff[BP];
    uint8_t *slut = (uint8_t *)lds + LDS_BYTES - 256;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long pix0 = (long)blockIdx.x * BP;
    const int oc0 = blockIdx.y * BN;
    const int hw = p.out_h * p.out_w;
    if (p.lut && tid < 64) ((uint32_t *)slut)[tid] = ((const uint32_t *)p.lut)[tid];
    fill_rowoff<BP>(p, rowoff, [=](int row) { long q = pix0 + row; return q < total_pix ? q : -1L; }, (unsigned)hw, dhw);

    const int cc = tid & 3;
    const int8_t *xbase[2];
    int iy0[2], ix0[2];
    bool rvalid[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        long pix = pix0 + (tid >> 2) + j * 64;
        rvalid[j] = pix < total_pix;
        long f = rvalid[j] ? pix / hw : 0;
        int rem = rvalid[j] ? (int)(pix - f * hw) : 0;
        int oy = rem / p.out_w, ox = rem - oy * p.out_w;
        iy0[j] = oy * p.stride_h - p.pad_top;
        ix0[j] = ox * p.stride_w - p.pad_left;
        xbase[j] = p.in + (size_t)f * p.in_stride;
    }
    int ky = 0, rc = cc * 16; // kernel row, byte inside the padded kernel row
    while (rc >= p.row_pad) { rc -= p.row_pad; ky++; }
    // K layout of a kernel row: kw taps of `ceff` bytes.  Small-channel layers are packed with every tap widened to 4 bytes
    // (mhip_conv_i8_pack_geom); this kernel serves those of them that neither conv_i8_rgb nor conv_i8_smallc takes
    const int ceff = mhip_small_c_dev(p.in_c, p.kw, p.out_c) ? 4 : p.in_c;
    const int nks = k64 / BK;
    constexpr int WLOADS = (BN * 4 + NTHREADS - 1) / NTHREADS;
    v4i xreg[2], wreg[WLOADS];

    auto load_global = [&](int ks) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            v4i v = {0, 0, 0, 0};
            const int iy = iy0[j] + ky;
            if (rvalid[j] && ky < p.kh && iy >= 0 && iy < p.in_h) {
                const int8_t *rowp = xbase[j] + ((long)iy * p.in_w + ix0[j]) * p.in_c;
                int8_t b[16];
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int rr = rc + e;
                    int8_t val = 0;
                    const int tap = rr / ceff, ch = rr - tap * ceff;
                    if (tap < p.kw && ch < p.in_c) {
                        const int ix = ix0[j] + tap;
                        if (ix >= 0 && ix < p.in_w) val = rowp[tap * p.in_c + ch];
                    }
                    b[e] = val;
                }
                v = *(v4i *)b;
            }
            xreg[j] = v;
        }
#pragma unroll
        for (int j = 0; j < WLOADS; j++) {
            const int idx = tid + j * NTHREADS;
            if (idx < BN * 4) wreg[j] = *(const v4i *)(p.w + (size_t)(oc0 + (idx >> 2)) * k64 + ks * BK + (idx & 3) * 16);
        }
        rc += BK;
        while (rc >= p.row_pad) { rc -= p.row_pad; ky++; }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; j++) *(v4i *)(lds + buf * STAGE + lds_off((tid >> 2) + j * 64, cc)) = xreg[j];
#pragma unroll
        for (int j = 0; j < WLOADS; j++) {
            const int idx = tid + j * NTHREADS;
            if (idx < BN * 4) *(v4i *)(lds + buf * STAGE + BP * BK + lds_off(idx >> 2, idx & 3)) = wreg[j];
        }
    };

    v4i acc[WOC][WPX];
    init_acc<WPX, WOC>(p, acc, oc0);

    load_global(0);
    store_lds(0);
    __syncthreads();
    const int frow = lane & 15, fchunk = lane >> 4;
    const int pxw = wv * 32;
    for (int ks = 0; ks < nks; ks++) {
        const int buf = ks & 1;
        if (ks + 1 < nks) load_global(ks + 1);
        v4i xb[WPX];
#pragma unroll
        for (int t = 0; t < WPX; t++) xb[t] = *(const v4i *)(lds + buf * STAGE + lds_off(pxw + t * 16 + frow, fchunk));
#pragma unroll
        for (int s = 0; s < WOC; s++) {
            v4i wa = *(const v4i *)(lds + buf * STAGE + BP * BK + lds_off(s * 16 + frow, fchunk));
#pragma unroll
            for (int t = 0; t < WPX; t++) acc[s][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, xb[t], acc[s][t], 0, 0, 0);
        }
        if (ks + 1 < nks) store_lds(buf ^ 1);
        __syncthreads();
    }
    epilogue<BP, BN, WPX, WOC>(p, acc, lds, slut, rowoff, oc0, pxw, 0, hw);
}

// packed weight / bias row that carries output channel `oc` (see epilogue_t): channels are permuted inside
// groups of G = 64 (oc_pad % 64 == 0, waves own 4 oc-subtiles) or 32 (2 subtiles)
extern "C" int mhip_conv_i8_oc_row(int oc, int oc_pad) {
    const int G = (oc_pad % 64 == 0) ? 64 : 32, NS = G / 16;
    const int g = oc / G, local = oc - g * G;
    const int q = local / (4 * NS), rem = local - q * 4 * NS, s = rem >> 2, r = rem & 3;
    return g * G + s * 16 + q * 4 + r;
}

extern "C" int mhip_conv_i8_is_safe(float cs) {
    // the accumulator is an int32, so |(float)acc| <= 2^31; with |cs| < 0.99 the product (and the
    // +-0.5) stays below 2^31 and is never NaN: the out-of-range branch of the reference is unreachable
    return cs == cs && cs < 0.99f && cs > -0.99f;
}

extern "C" int mhip_conv_i8_lut2_ok(float cs) {
    // round-half-away(x) == f(trunc(2x)) for every float except |x| = 0x3EFFFFFF (0.49999997), which the reference's
    // float add rounds up to 1: refuse the half-step table when some int32 accumulator lands exactly there
    if (!mhip_conv_i8_is_safe(cs)) return 0;
    const float a = cs < 0 ? -cs : cs;
    if (!(a >= 1e-6f)) return 0; // tiny scales: many accumulators crowd around any given product
    const float quirk = 0.49999997f;
    const double a0 = (double)quirk / (double)a;
    for (int d = -3; d <= 3; d++) {
        const double c = (double)(long long)(a0 + 0.5) + d;
        if (c < 1.0 || c > 2147483647.0) continue;
        volatile float prod = (float)(int)c * a;
        if (prod == quirk) return 0;
    }
    return 1;
}

extern "C" int mhip_conv_i8_small_c(int in_c, int kw, int out_c) { return in_c <= 4 && kw <= 8 && out_c <= 64; }

extern "C" void mhip_conv_i8_pack_geom(int in_c, int kw, int out_c, int *row_pad, int *oc_pad, int *c_eff) {
    const int small = mhip_conv_i8_small_c(in_c, kw, out_c);
    if (row_pad) *row_pad = small ? 32 : (kw * in_c + 15) & ~15; // small: every pixel widened to 4 bytes
    if (oc_pad) *oc_pad = (out_c + 31) & ~31;
    if (c_eff) *c_eff = small ? 4 : in_c;
}

static tune_t g_tune;
static int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
const tune_t &conv_i8_tune_state() {
    if (!g_tune.init) {
        g_tune.persist = env_int("MARS_HIP_PERSIST", 1);
        g_tune.persist_stages = env_int("MARS_HIP_PSTAGES", 2);
        g_tune.persist_maxk = env_int("MARS_HIP_PERSIST_MAXK", 8); // measured: deeper K loops gain nothing from walking tiles
        g_tune.persist_slots = env_int("MARS_HIP_PSLOTS", 0);
        g_tune.stages = env_int("MARS_HIP_STAGES", 0);
        g_tune.bpx = env_int("MARS_HIP_BPX", 0);
        g_tune.variant = env_int("MARS_HIP_VARIANT", 0);
        g_tune.bufmode = env_int("MARS_HIP_BUFMODE", 1);
        g_tune.wres = env_int("MARS_HIP_WRES", 3);
        g_tune.rgb_direct = env_int("MARS_HIP_RGB_DIRECT", 1);
        g_tune.small_batch = env_int("MARS_HIP_SMALL_BATCH", 1);
        g_tune.rows = env_int("MARS_HIP_ROWS", 1);
        g_tune.few_wgs = env_int("MARS_HIP_FEW_WGS", 256);
        g_tune.patch_ring = env_int("MARS_HIP_PATCH_RING", 0);
        g_tune.patch_lds_kb = env_int("MARS_HIP_PATCH_LDS_KB", 80);
        g_tune.init = 1;
    }
    return g_tune;
}
static inline const tune_t &tune() { return conv_i8_tune_state(); }
// value == NULL-safe accessor pair: set (get == nullptr) or read (get != nullptr) one launch-policy knob
static int tune_access(const char *key, int value, int *get) {
    (void)conv_i8_tune_state();
    struct { const char *k; int *v; } tab[] = {{"persist", &g_tune.persist}, {"persist_stages", &g_tune.persist_stages},
                                               {"persist_maxk", &g_tune.persist_maxk}, {"persist_slots", &g_tune.persist_slots}, {"wres", &g_tune.wres}, {"rgb_direct", &g_tune.rgb_direct}, {"small_batch", &g_tune.small_batch},
                                               {"rows", &g_tune.rows}, {"few_wgs", &g_tune.few_wgs}, {"patch_ring", &g_tune.patch_ring}, {"patch_lds_kb", &g_tune.patch_lds_kb}, {"stages", &g_tune.stages}, {"bpx", &g_tune.bpx}, {"variant", &g_tune.variant}, {"bufmode", &g_tune.bufmode}};
    for (auto &e : tab)
        if (key && !strcmp(key, e.k)) {
            if (get) *get = *e.v;
            else *e.v = value;
            return 0;
        }
    return -1;
}
extern "C" int mhip_conv_i8_tune(const char *key, int value) { return tune_access(key, value, nullptr); }
extern "C" int mhip_conv_i8_tune_get(const char *key, int *value) { return value ? tune_access(key, 0, value) : -1; }

// buffer-addressed K loop: a 64-byte step inside one tap (in_c >= 64, power of two), 31-bit offsets, tap masks
static int buf_mode(const mhip_conv_i8_t *p, int k64) {
    if (!tune().bufmode) return 0;
    return p->in_c >= 64 && (p->in_c & (p->in_c - 1)) == 0 && p->kh * p->kw <= 32 && (long)p->kh * p->kw * (p->kw - 1) < 65536 &&
           in_extent_bytes(p) <= 0x7fffffffL && (long)p->oc_pad * k64 <= 0x7fffffffL;
}

template <int BPX, int BN, int STAGES, int KS = 1, int NW = 4>
static int launch_mfma(const mhip_conv_i8_t *p, long total_pix, int k64) {
    const unsigned npt = (unsigned)((total_pix + BPX - 1) / BPX), noc = (unsigned)(p->oc_pad / BN);
    const unsigned nblk = npt * noc;
    const int nks = k64 / BK, nst = nks / KS, used = nst < STAGES ? nst : STAGES;
    if (KS > 1 && (nks % KS) != 0) return -1;
    size_t ring = (size_t)used * KS * (BPX + BN) * BK, tile = (size_t)BPX * (BN + OPAD);
    const size_t lds = BPX * 8 + LUTB + (ring > tile ? ring : tile);
    if (lds > 64 * 1024) {
        static bool attr = false;
        if (!attr && hipFuncSetAttribute((const void *)conv_i8_mfma<BPX, BN, STAGES, KS, NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(BPX * 8 + LUTB + (size_t)STAGES * KS * (BPX + BN) * BK)) != hipSuccess)
            return mhip_check(hipErrorUnknown, "conv_i8_mfma LDS attribute");
        attr = true;
    }
    // in_c a power of two and tap/kw small enough for the 16-bit reciprocal: shift-based K position
    int lg = -1;
    if ((p->in_c & (p->in_c - 1)) == 0 && (long)p->kh * p->kw * (p->kw - 1) < 65536) {
        lg = 0;
        while ((1 << lg) < p->in_c) lg++;
    }
    const unsigned magic = ((65536u + (unsigned)p->kw - 1u) / (unsigned)p->kw);
    hipLaunchKernelGGL((conv_i8_mfma<BPX, BN, STAGES, KS, NW>), dim3(nblk), dim3(NW * 64), lds, mhip_stream_native(), *p, total_pix,
                       k64, (const int8_t *)mhip_zero_page(), noc, nblk, lg, magic,
                       make_fastdiv((unsigned)(p->out_h * p->out_w)), make_fastdiv((unsigned)p->out_w),
                       lg >= 0 ? buf_mode(p, k64) : 0, (unsigned)in_extent_bytes(p));
    return mhip_check(hipGetLastError(), "conv_i8_mfma launch");
}


// LDS of the weights-resident form: LUT + 2 pixel-tile stages + every K step of one channel tile
template <int BPX, int BN>
static size_t wres_lds(int k64) { return LUTB + 2 * (size_t)BPX * BK + (size_t)(k64 / BK) * BN * BK; }

template <int BPX, int BN, int STAGES, bool HAS_LUT, bool SEG = false, bool PAIR = false>
static int launch_persist_t(const mhip_conv_i8_t *p, long total_pix, int k64, int lg, unsigned magic,
                            const mhip_conv_i8_t *second = nullptr, int wres = 0) {
    if (BN == 16 && (p->out_c > 16 || p->oc_pad != 32 || (PAIR && (second->out_c > 16 || second->oc_pad != 32)))) return -1; // (one tile of 16 channels out of a 32-row image)
    const unsigned noc0 = BN == 16 ? 1u : (unsigned)(p->oc_pad / BN);
    const unsigned npt = (unsigned)((total_pix + BPX - 1) / BPX), noc = noc0 + (PAIR ? (BN == 16 ? 1u : (unsigned)(second->oc_pad / BN)) : 0u);
    conv_out_side_t alt;
    memset(&alt, 0, sizeof(alt));
    if (PAIR) {
        alt.out = second->out; alt.out_stride = second->out_stride; alt.w = second->w; alt.bias = second->bias;
        alt.lut = second->lut; alt.lut2 = second->lut2; alt.out_c = second->out_c; alt.relu = second->relu; alt.out_pix_stride = second->out_pix_stride;
        alt.out_ch_off = second->out_ch_off; alt.cs = second->cs;
    }
    if (PAIR) alt.out_bytes = (unsigned)persist_out_bytes(second);
    if (wres && STAGES != 2) return -1;
    const size_t lds = wres ? wres_lds<BPX, BN>(k64) : LUTB + (size_t)STAGES * (BPX + BN) * BK;
    if (wres) { // its own occupancy (LDS differs per layer) and longer runs: the weight fetch must amortise
        static bool attr = false;
        if (!attr && hipFuncSetAttribute((const void *)conv_i8_persist<BPX, BN, STAGES, HAS_LUT, SEG, PAIR>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)
            return mhip_check(hipErrorUnknown, "conv_i8_persist LDS attribute");
        attr = true;
        int occ = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, conv_i8_persist<BPX, BN, STAGES, HAS_LUT, SEG, PAIR>, NTHREADS, lds) != hipSuccess ||
            hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return mhip_check(hipErrorUnknown, "conv_i8_persist occupancy query");
        unsigned g = (unsigned)((occ > 0 ? occ : 1) * prop.multiProcessorCount) / noc;
        if (tune().persist_slots > 0) g = (unsigned)tune().persist_slots / noc;
        if (g < 1) g = 1;
        if (g > npt) g = npt;
        hipLaunchKernelGGL((conv_i8_persist<BPX, BN, STAGES, HAS_LUT, SEG, PAIR>), dim3(noc * g), dim3(NTHREADS), lds,
                           mhip_stream_native(), *p, (unsigned)total_pix, k64, (const int8_t *)mhip_zero_page(), noc, npt, g, lg,
                           magic, make_fastdiv((unsigned)(p->out_h * p->out_w)), make_fastdiv((unsigned)p->out_w),
                           (unsigned)persist_out_bytes(p), alt, noc0, SEG ? 0 : buf_mode(p, k64), (unsigned)in_extent_bytes(p), 1);
        return mhip_check(hipGetLastError(), "conv_i8_persist (resident weights) launch");
    }
    static int slots = 0; // workgroups of this instantiation the device holds at once
    if (!slots) {
        int occ = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, conv_i8_persist<BPX, BN, STAGES, HAS_LUT, SEG, PAIR>, NTHREADS, lds) != hipSuccess ||
            hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return mhip_check(hipErrorUnknown, "conv_i8_persist occupancy query");
        slots = (occ > 0 ? occ : 1) * prop.multiProcessorCount;
    }
    // 4 workgroups per device slot: short enough runs that a workgroup which has to wait for a slot (the detection
    // tail of the previous batch shares the CUs) costs little, long enough to keep the cross-tile prefetch (measured)
    unsigned ngrp = (unsigned)(tune().persist_slots > 0 ? tune().persist_slots : 4 * slots) / noc;
    if (ngrp < 1) ngrp = 1;
    if (ngrp > npt) ngrp = npt;
    hipLaunchKernelGGL((conv_i8_persist<BPX, BN, STAGES, HAS_LUT, SEG, PAIR>), dim3(noc * ngrp), dim3(NTHREADS), lds,
                       mhip_stream_native(), *p, (unsigned)total_pix, k64, (const int8_t *)mhip_zero_page(), noc, npt, ngrp, lg,
                       magic, make_fastdiv((unsigned)(p->out_h * p->out_w)), make_fastdiv((unsigned)p->out_w),
                       (unsigned)persist_out_bytes(p), alt, noc0, SEG ? 0 : buf_mode(p, k64), (unsigned)in_extent_bytes(p), 0);
    return mhip_check(hipGetLastError(), "conv_i8_persist launch");
}

// 1 = launched (rc in *rc), 0 = this layer is not eligible for the persistent kernel
// ---- launch variants of the in_c % 16 == 0 kernels.  A variant = (form, pixels per workgroup, ring depth); every
// variant computes the same bytes.  p->variant == 0 takes the measured default policy below; the host can pin a
// variant per layer after timing the candidates on the device (mars_hip_autotune).
//   code = 1 + persist + 2*(bpx == 256) + 4*(stages == 3)
//   code = 9 / 10 / 11: patch-staged kernel with 8 / 16 / 4 tile rows
//   code = 12: one tile per workgroup, 128 pixels, 2 ring stages of 128 K bytes each (even number of K steps)
//   code = 13: one 256 x 128 tile per 8-wave workgroup, 3 stages
//   code = 14 / 15: tile walker with the weights of its channel tile resident in LDS, 128 / 256 pixels
//   code = 18 / 19: 128-byte K steps (whole-line DMA requests), 128 x 128 tile on 4 waves / 256 x 128 tile on 8 waves
//   code = 20: conv_i8_rows (round 4): whole-row tiles, patch-staged input, streamed weights, persistent, deep 3x3 stride 1
#define NVARIANTS 20 // 16, 17 were measured-and-dropped forms (round 2), 6 / 8 = the three-stage tile walker (pruned in round 4)
struct variant_t {
    int persist, bpx, stages, patch, ks2, w8, wres, r128, rows;
};
static int variant_code(const variant_t &v) {
    if (v.rows) return 20;
    if (v.r128) return 17 + v.r128;
    if (v.wres) return v.bpx == 256 ? 15 : 14;
    if (v.w8) return 13;
    if (v.ks2) return 12;
    if (v.patch) return v.patch == 16 ? 10 : (v.patch == 8 ? 9 : 11);
    return 1 + (v.persist ? 1 : 0) + (v.bpx == 256 ? 2 : 0) + (v.stages == 3 ? 4 : 0);
}
static variant_t variant_of(int code) {
    if (code == 20) return variant_t{0, 0, 0, 0, 0, 0, 0, 0, 1};
    if (code == 18 || code == 19) return variant_t{0, 0, 0, 0, 0, 0, 0, code - 17, 0};
    if (code == 14 || code == 15) return variant_t{1, code == 15 ? 256 : 128, 2, 0, 0, 0, 1, 0, 0};
    if (code == 13) return variant_t{0, 256, 3, 0, 0, 1, 0, 0, 0};
    if (code == 12) return variant_t{0, 128, 2, 0, 1, 0, 0, 0, 0};
    if (code >= 9 && code <= 11) return variant_t{0, 0, 0, code == 10 ? 16 : (code == 9 ? 8 : 4), 0, 0, 0, 0, 0};
    if (code > 8 || code == 6 || code == 8) return variant_t{-1, 0, 0, 0, 0, 0, 0, 0, 0}; // 6, 8, 16, 17: retired codes
    const int c = code - 1;
    return variant_t{c & 1, (c & 2) ? 256 : 128, (c & 4) ? 3 : 2, 0, 0, 0, 0, 0, 0};
}

// ---- 128-byte K steps (conv_i8_r128): eligibility, launch
static bool r128_ok(const mhip_conv_i8_t *p) {
    const int C = p->in_c;
    return C >= 128 && (C & (C - 1)) == 0 && p->oc_pad % 128 == 0 && p->kh * p->kw <= 32 && (long)p->kh * p->kw * (p->kw - 1) < 65536 &&
           p->row_pad == p->kw * C && p->nseg <= 1 && in_extent_bytes(p) <= 0x7fffffffL && (long)p->oc_pad * p->kh * p->row_pad <= 0x7fffffffL;
}
template <int BPX, int BN, int NW>
static int launch_r128_t(const mhip_conv_i8_t *p, long total_pix) {
    const int k128 = p->kh * p->row_pad; // taps * C: a multiple of 128
    const unsigned npt = (unsigned)((total_pix + BPX - 1) / BPX), noc = (unsigned)(p->oc_pad / BN);
    size_t ring = 2 * (size_t)(BPX + BN) * 128, tile = (size_t)BPX * (BN + OPAD);
    const size_t lds = BPX * 8 + LUTB + (ring > tile ? ring : tile);
    static bool attr = false;
    if (!attr && hipFuncSetAttribute((const void *)conv_i8_r128<BPX, BN, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return mhip_check(hipErrorUnknown, "conv_i8_r128 LDS attribute");
    attr = true;
    int lg = 0;
    while ((1 << lg) < p->in_c) lg++;
    const unsigned magic = ((65536u + (unsigned)p->kw - 1u) / (unsigned)p->kw);
    hipLaunchKernelGGL((conv_i8_r128<BPX, BN, NW>), dim3(npt * noc), dim3(NW * 64), lds, mhip_stream_native(), *p, total_pix, k128, noc,
                       npt * noc, lg, magic, make_fastdiv((unsigned)(p->out_h * p->out_w)), make_fastdiv((unsigned)p->out_w),
                       (unsigned)in_extent_bytes(p));
    return mhip_check(hipGetLastError(), "conv_i8_r128 launch");
}
static int launch_r128(const mhip_conv_i8_t *p, long total_pix, int form) {
    if (!r128_ok(p)) return -1;
    return form == 2 ? launch_r128_t<256, 128, 8>(p, total_pix) : launch_r128_t<128, 128, 4>(p, total_pix);
}
// a convolution whose input is a virtual concatenation: 1x1, stride 1, unpadded, segments tile [0, in_c) in steps of 32
static bool seg_valid(const mhip_conv_i8_t *p) {
    if (p->nseg < 2 || p->nseg > 4 || p->kh != 1 || p->kw != 1 || p->stride_h != 1 || p->stride_w != 1 || p->pad_top ||
        p->pad_left || p->in_h != p->out_h || p->in_w != p->out_w)
        return false;
    int c = 0;
    if (p->seg_up && ((p->out_h | p->out_w) & 1)) return false;
    if (p->seg_up >> p->nseg) return false;
    for (int i = 0; i < 4; i++) {
        if (i < p->nseg) {
            if (!p->seg_in[i] || p->seg_c[i] <= 0 || (p->seg_c[i] & 31) || p->seg_c0[i] != c) return false;
            c += p->seg_c[i];
        } else if (p->seg_c0[i] != 0x7fffffff) {
            return false;
        }
    }
    return c == p->in_c;
}
static bool persist_eligible(const mhip_conv_i8_t *p) {
    if (p->add) return false; // the fused residual Add lives in the one-tile and patch-staged epilogues
    return !p->out_nchw && p->safe && (p->in_c & (p->in_c - 1)) == 0 && p->kh * p->kw <= 32 &&
           (long)p->kh * p->kw * (p->kw - 1) < 65536 && persist_out_bytes(p) <= 0x7fffffffL;
}
static bool wres_default(const mhip_conv_i8_t *p, int nks, int *bpx) {
    if (!tune().wres || p->kh * p->kw != 1 || p->in_c < 64 || p->out_h * p->out_w > 6400) return false;
    const int bn = p->oc_pad % 128 == 0 ? 128 : (p->oc_pad % 64 == 0 ? 64 : 32);
    const int px = (bn == 64 && p->out_h * p->out_w >= 6400) ? 256 : 128;
    if (LUTB + 2 * (size_t)px * BK + (size_t)nks * bn * BK > 80 * 1024) return false;
    *bpx = px;
    return true;
}

static variant_t default_variant(const mhip_conv_i8_t *p, int nks) {
    variant_t v;
    v.patch = 0;
    v.ks2 = 0;
    v.w8 = 0;
    v.wres = 0;
    v.r128 = 0;
    v.rows = 0;
    // wide, shallow k x k layers: the patch-staged kernel wins wherever its double-buffered form fits (measured on the
    // 160x160 and 80x80 layers of yolov5s: 1.25-1.9x over the implicit-GEMM forms)
    int ring = 0;
    // Few frames: a launch whose large-batch tiling yields fewer workgroups than the device has CUs is bound by the
    // latency of ONE workgroup's K walk, not by bytes per MAC.  Smaller tiles put more CUs to work and shorten every
    // step: 4-row patches, the 128 x 128 tile with 128-byte K steps for deep K loops (half the barriers per MAC, four
    // waves), the plain two-stage tile walker without the up-front weight fetch for 1 x 1 layers.  These are the
    // autotuner's choices at batch 1 (yolov5s twin, 640 x 640: 0.62 -> 0.53 ms per frame).
    const long wg_large = ((long)p->frames * p->out_h * p->out_w + 255) / 256 * ((p->oc_pad + 127) / 128);
    const bool few = wg_large < tune().few_wgs && tune().persist && !tune().bpx && !tune().stages && tune().small_batch;
    if (few) {
        if (conv_i8_patch_ok(p, 4, &ring)) {
            v.persist = 0; v.bpx = 0; v.stages = 0; v.patch = 4;
            return v;
        }
        if (nks >= 8 && r128_ok(p)) {
            v.persist = 0; v.bpx = 0; v.stages = 0; v.r128 = 1;
            return v;
        }
        v.bpx = 128;
        v.persist = nks <= tune().persist_maxk && persist_eligible(p);
        if (((p->out_c | p->out_pix_stride | p->out_ch_off) & 15) != 0 && p->nseg <= 1) v.persist = 0;
        v.stages = v.persist ? 2 : 3;
        return v;
    }
    // deep 3x3 stride-1 layers on 20-wide maps with 256+ input channels: the persistent whole-row form (conv_i8_rows) --
    // measured 70.8 vs 77.7 us (3x3 256->256 @20x20) and 204 vs 252 us (512->512) at batch 256; on 40-wide maps with 128
    // channels (two chunks per tile: the in-line epilogue is a fifth of a tile) it loses, 81.5 vs 79.2 us
    if (tune().persist && !tune().bpx && !tune().stages && tune().rows && p->out_w <= 20 && p->in_c >= 256 && conv_i8_rows_ok(p)) {
        v.persist = 0; v.bpx = 0; v.stages = 0; v.rows = 1;
        return v;
    }
    if (tune().persist && !tune().bpx && !tune().stages) {
        // 16 output rows per workgroup wherever that patch fits at all (single-buffered included: the taller patch
        // re-reads fewer halo rows, measured 5-40 % over 8 rows on the 160x160 / 80x80 layers), else 8 rows
        if (conv_i8_patch_ok(p, 16, &ring)) {
            v.persist = 0; v.bpx = 0; v.stages = 0; v.patch = 16;
            return v;
        }
        if (conv_i8_patch_ok(p, 8, &ring) && ring >= 2) {
            v.persist = 0; v.bpx = 0; v.stages = 0; v.patch = 8;
            return v;
        }
    }
    // pixels per workgroup: 256 halves the weight-tile traffic and per-workgroup overhead of the narrow, shallow
    // configurations; 128 keeps one more workgroup per CU everywhere else
    v.bpx = tune().bpx ? tune().bpx : ((p->oc_pad % 128 != 0 && nks <= 2) ? 256 : 128);
    v.persist = tune().persist && nks <= tune().persist_maxk && persist_eligible(p);
    // rows that are not 16-byte aligned (the 255-channel heads): the one-tile form with its LDS-staged copy-out
    // beats the ragged buffer stores of the tile walker (measured 233 vs 298 us on the 80x80 head)
    if (((p->out_c | p->out_pix_stride | p->out_ch_off) & 15) != 0 && p->nseg <= 1) v.persist = 0;
    // ring depth: 3 stages beat 4 everywhere (occupancy > depth); the tile-walking form is best with 2
    v.stages = v.persist ? 2 : (tune().stages ? tune().stages : (nks <= 4 ? 2 : 3));
    if (v.stages != 3) v.stages = 2;
    // 1x1 layers of 64+ input channels on maps up to 80x80: the tile walker with its weights resident in LDS (the ring
    // carries pixel tiles only -- half the LDS-DMA bytes; measured 5-25 % faster on every such layer of yolov5s, and
    // no gain on the 160x160 maps)
    if (v.persist && v.stages == 2 && !tune().bpx && wres_default(p, nks, &v.bpx)) v.wres = 1;
    // deep K loops with 128-channel tiles: the 8-wave 256 x 128 tile moves 25 % fewer DMA bytes per MAC (measured
    // 3-15 % faster on every such layer of yolov5s)
    if ((!v.persist || (!v.wres && nks >= 8 && p->nseg <= 1)) && !tune().bpx && !tune().stages && p->oc_pad % 128 == 0 &&
        nks >= (tune().persist_maxk < 8 ? tune().persist_maxk + 1 : 8)) {
        v.persist = 0; v.w8 = 1; v.bpx = 256; v.stages = 3;
    }
    // never-materialised concat inputs (tile walker only) with 8+ K steps: 256 pixels per workgroup (measured 5-8 %)
    if (p->nseg > 1 && !v.wres && !tune().bpx && nks >= 8) v.bpx = 256;
    return v;
}

template <int BPX, int BN>
static int launch_variant_t(const mhip_conv_i8_t *p, long total_pix, int k64, const variant_t &v) {
    if (v.persist) {
        if (!persist_eligible(p)) return -1;
        int lg = 0;
        while ((1 << lg) < p->in_c) lg++;
        const unsigned magic = ((65536u + (unsigned)p->kw - 1u) / (unsigned)p->kw);
        if (v.wres && wres_lds<BPX, BN>(k64) > 80 * 1024) return -1;
        if (p->nseg > 1) // never-materialised concat input: the two-stage tile walker only
            return p->lut ? launch_persist_t<BPX, BN, 2, true, true>(p, total_pix, k64, lg, magic, nullptr, v.wres)
                          : launch_persist_t<BPX, BN, 2, false, true>(p, total_pix, k64, lg, magic, nullptr, v.wres);
        if (v.wres)
            return p->lut ? launch_persist_t<BPX, BN, 2, true>(p, total_pix, k64, lg, magic, nullptr, 1)
                          : launch_persist_t<BPX, BN, 2, false>(p, total_pix, k64, lg, magic, nullptr, 1);
        if (v.stages == 2)
            return p->lut ? launch_persist_t<BPX, BN, 2, true>(p, total_pix, k64, lg, magic)
                          : launch_persist_t<BPX, BN, 2, false>(p, total_pix, k64, lg, magic);
        return -1; // the three-stage tile walker (variants 6 / 8) is retired: round 4 pruning, never picked by the policy or the tuner
    }
    if constexpr (BN == 16) {
        return -1; // (the 16-channel tile exists in the tile walker only)
    } else {
        if (v.w8) return BN == 128 ? launch_mfma<256, 128, 3, 1, 8>(p, total_pix, k64) : -1;
        if (v.ks2) return BPX == 128 && BN >= 64 ? launch_mfma<128, (BN >= 64 ? BN : 64), 2, 2>(p, total_pix, k64) : -1;
        return v.stages == 2 ? launch_mfma<BPX, BN, 2>(p, total_pix, k64) : launch_mfma<BPX, BN, 3>(p, total_pix, k64);
    }
}
// out_c <= 16 in a 32-row weight image: the tile walker runs one 16-channel tile (conv_i8_persist<.., 16, ..>); MARS_HIP_NO_BN16 = the 32-row tile (A / B)
static bool bn16_ok(const mhip_conv_i8_t *p) {
    static const bool off = getenv("MARS_HIP_NO_BN16") != nullptr;
    return !off && p->out_c <= 16 && p->oc_pad == 32 && p->nseg <= 1;
}

static int launch_variant(const mhip_conv_i8_t *p, long total_pix, int k64, const variant_t &v) {
    if (v.persist < 0) return -1; // a retired variant code
    if (v.rows) return conv_i8_launch_rows(p);
    if (v.r128) return launch_r128(p, total_pix, v.r128);
    if (v.patch) return conv_i8_launch_patch(p, k64, v.patch);
    const int bn = p->oc_pad % 128 == 0 ? 128 : (p->oc_pad % 64 == 0 ? 64 : 32);
    const bool b16 = v.persist && v.stages == 2 && bn16_ok(p);
    if (v.bpx == 256) {
        if (bn == 128) return launch_variant_t<256, 128>(p, total_pix, k64, v);
        if (bn == 64) return launch_variant_t<256, 64>(p, total_pix, k64, v);
        if (b16) return launch_variant_t<256, 16>(p, total_pix, k64, v);
        return launch_variant_t<256, 32>(p, total_pix, k64, v);
    }
    if (bn == 128) return launch_variant_t<128, 128>(p, total_pix, k64, v);
    if (bn == 64) return launch_variant_t<128, 64>(p, total_pix, k64, v);
    if (b16) return launch_variant_t<128, 16>(p, total_pix, k64, v);
    return launch_variant_t<128, 32>(p, total_pix, k64, v);
}

// Two convolutions over the same input in one launch (conv_i8_persist<PAIR>).  -2 = the pair is not eligible (the
// caller launches them one after the other), otherwise the launch's return code.
template <int BPX, int BN>
static int launch_pair_t(const mhip_conv_i8_t *a, const mhip_conv_i8_t *b, long total_pix, int k64, int lg, unsigned magic,
                         int wres) {
    if (a->nseg > 1) return -2; // pairs over a never-materialised concat measured slower than two tuned launches: not built (round 4)
    return launch_persist_t<BPX, BN, 2, true, false, true>(a, total_pix, k64, lg, magic, b, wres);
}
extern "C" int mhip_conv_i8_pair(const mhip_conv_i8_t *a, const mhip_conv_i8_t *b) {
    if (!a || !b || !tune().persist || tune().variant) return -2;
    // identical input side and geometry; both with a fused LUT; both eligible for the tile-walking form
    if (a->in != b->in || a->in_stride != b->in_stride || a->frames != b->frames || a->in_h != b->in_h || a->in_w != b->in_w ||
        a->in_c != b->in_c || a->out_h != b->out_h || a->out_w != b->out_w || a->kh != b->kh || a->kw != b->kw ||
        a->stride_h != b->stride_h || a->stride_w != b->stride_w || a->pad_top != b->pad_top || a->pad_left != b->pad_left ||
        a->row_pad != b->row_pad || a->nseg != b->nseg || a->seg_up != b->seg_up || !a->lut || !b->lut || a->add || b->add ||
        a->out_nchw || b->out_nchw)
        return -2;
    for (int i = 0; i < a->nseg; i++)
        if (a->seg_in[i] != b->seg_in[i] || a->seg_c[i] != b->seg_c[i] || a->seg_stride[i] != b->seg_stride[i]) return -2;
    if ((a->in_c % 16) != 0 || mhip_conv_i8_small_c(a->in_c, a->kw, a->out_c) || mhip_conv_i8_small_c(b->in_c, b->kw, b->out_c))
        return -2;
    if (a->nseg > 1 && (!seg_valid(a) || !seg_valid(b))) return -2;
    if (!persist_eligible(a) || !persist_eligible(b)) return -2;
    const int bn = a->oc_pad % 128 == 0 ? 128 : (a->oc_pad % 64 == 0 ? 64 : 32);
    const int bnb = b->oc_pad % 128 == 0 ? 128 : (b->oc_pad % 64 == 0 ? 64 : 32);
    if (bn != bnb || !mhip_zero_page()) return -2;
    const long total_pix = (long)a->frames * a->out_h * a->out_w;
    const int k64 = (a->kh * a->row_pad + BK - 1) / BK * BK, nks = k64 / BK;
    if (total_pix <= 0 || total_pix > 0x7fffffffL) return -2;
    int lg = 0;
    while ((1 << lg) < a->in_c) lg++;
    const unsigned magic = ((65536u + (unsigned)a->kw - 1u) / (unsigned)a->kw);
    const variant_t dv = default_variant(a, nks);
    const int bpx = dv.bpx ? dv.bpx : 128, wres = dv.persist && dv.wres && (tune().wres & 2);
    const bool b16 = bn16_ok(a) && bn16_ok(b);
    if (bpx == 256) {
        if (bn == 128) return launch_pair_t<256, 128>(a, b, total_pix, k64, lg, magic, wres);
        if (bn == 64) return launch_pair_t<256, 64>(a, b, total_pix, k64, lg, magic, wres);
        if (b16) return launch_pair_t<256, 16>(a, b, total_pix, k64, lg, magic, wres);
        return launch_pair_t<256, 32>(a, b, total_pix, k64, lg, magic, wres);
    }
    if (bn == 128) return launch_pair_t<128, 128>(a, b, total_pix, k64, lg, magic, wres);
    if (bn == 64) return launch_pair_t<128, 64>(a, b, total_pix, k64, lg, magic, wres);
    if (b16) return launch_pair_t<128, 16>(a, b, total_pix, k64, lg, magic, wres);
    return launch_pair_t<128, 32>(a, b, total_pix, k64, lg, magic, wres);
}

extern "C" int mhip_conv_i8_seg_ok(const mhip_conv_i8_t *p) {
    return p && (p->in_c % 16) == 0 && !mhip_conv_i8_small_c(p->in_c, p->kw, p->out_c) && seg_valid(p) && persist_eligible(p);
}

// candidate variant codes of a layer (0 when the layer is not served by these kernels), default first
extern "C" int mhip_conv_i8_variants(const mhip_conv_i8_t *p, int *codes, int max) {
    if (!p || (p->in_c % 16) != 0 || mhip_conv_i8_small_c(p->in_c, p->kw, p->out_c)) return 0;
    const int k64 = (p->kh * p->row_pad + BK - 1) / BK * BK, nks = k64 / BK;
    int n = 0;
    if (p->pre_w) { // fused bottleneck: the patch-staged kernel at 16 / 8 / 4 tile rows
        const int first = conv_i8_pre_tile_rows(p);
        for (int th : {first, 16, 8, 4})
            if (th && n < max && conv_i8_patch_ok(p, th, nullptr)) {
                const int code = th == 16 ? 10 : (th == 8 ? 9 : 11);
                bool seen = false;
                for (int i = 0; i < n; i++) seen |= codes[i] == code;
                if (!seen) codes[n++] = code;
            }
        return n;
    }
    if (p->nseg > 1) { // the tile walker only: plain, or with resident weights where they fit
        const int bn = p->oc_pad % 128 == 0 ? 128 : (p->oc_pad % 64 == 0 ? 64 : 32);
        const variant_t dv = default_variant(p, nks);
        const int dflt = dv.persist && dv.wres ? (dv.bpx == 256 ? 15 : 14) : (dv.bpx == 256 ? 4 : 2);
        if (n < max) codes[n++] = dflt;
        if (n < max && dflt != 2) codes[n++] = 2;
        if (n < max && dflt != 4) codes[n++] = 4;
        if (n < max && dflt != 14 && LUTB + 2 * (size_t)128 * BK + (size_t)nks * bn * BK <= 80 * 1024) codes[n++] = 14;
        if (n < max && dflt != 15 && LUTB + 2 * (size_t)256 * BK + (size_t)nks * bn * BK <= 80 * 1024) codes[n++] = 15;
        return n;
    }
    const int dflt = variant_code(default_variant(p, nks));
    if (n < max) codes[n++] = dflt;
    for (int code = 1; code <= NVARIANTS; code++) {
        const variant_t v = variant_of(code);
        if (code == dflt) continue;
        if (v.patch && !conv_i8_patch_ok(p, v.patch, nullptr)) continue;
        if (v.persist && !persist_eligible(p)) continue;
        if (v.ks2 && ((nks & 1) || nks < 4 || p->oc_pad % 64 != 0)) continue;
        if (v.w8 && (p->oc_pad % 128 != 0 || nks < 3)) continue;
        if (v.persist < 0) continue; // retired codes
        if (v.r128 && !r128_ok(p)) continue;
        if (v.rows && !conv_i8_rows_ok(p)) continue;
        if (v.wres) {
            const int bn = p->oc_pad % 128 == 0 ? 128 : (p->oc_pad % 64 == 0 ? 64 : 32);
            if (LUTB + 2 * (size_t)v.bpx * BK + (size_t)nks * bn * BK > 80 * 1024) continue;
        }
        if (!v.persist && v.stages == 3 && nks <= 2) continue; // identical to the 2-stage launch
        if (n < max) codes[n++] = code;
    }
    return n;
}

template <int BN>
static int launch_generic(const mhip_conv_i8_t *p, long total_pix, int k64) {
    dim3 grid((unsigned)((total_pix + BP - 1) / BP), (unsigned)(p->oc_pad / BN));
    hipLaunchKernelGGL((conv_i8_generic<BN>), grid, dim3(NTHREADS), 0, mhip_stream_native(), *p, total_pix, k64,
                       make_fastdiv((unsigned)(p->out_h * p->out_w)));
    return mhip_check(hipGetLastError(), "conv_i8_generic launch");
}

extern "C" int mhip_conv_i8(const mhip_conv_i8_t *p) {
    // host-side shape checks: the kernels trust these (a faulting kernel can reset the node)
    if (!p || !p->in || !p->out || !p->w) return -1;
    if (p->out_pix_stride < 0 || p->out_ch_off < 0 || (p->out_nchw && (p->out_pix_stride || p->out_ch_off)) ||
        (p->out_pix_stride && p->out_pix_stride < p->out_ch_off + p->out_c))
        return -1;
    if (p->frames <= 0 || p->in_h <= 0 || p->in_w <= 0 || p->in_c <= 0 || p->out_h <= 0 || p->out_w <= 0 ||
        p->out_c <= 0 || p->kh <= 0 || p->kw <= 0 || p->stride_h < 0 || p->stride_w < 0)
        return -1;
    int row_pad, oc_pad;
    mhip_conv_i8_pack_geom(p->in_c, p->kw, p->out_c, &row_pad, &oc_pad, nullptr);
    if (row_pad != p->row_pad || oc_pad != p->oc_pad) return -1;
    if (p->add && (!p->safe || p->out_nchw || ((p->out_c | p->out_pix_stride | p->out_ch_off) & 15) || (p->in_c % 16) != 0 ||
                   mhip_conv_i8_small_c(p->in_c, p->kw, p->out_c) || p->nseg > 1))
        return -1;
    const long total_pix = (long)p->frames * p->out_h * p->out_w;
    const int k64 = (p->kh * p->row_pad + BK - 1) / BK * BK;
    if (total_pix <= 0 || total_pix > 0x7fffffffL || (total_pix + BP - 1) / BP * (oc_pad / 32) > 0x7fffffffL) return -1;
    if (p->in_planar) { // planes [c][H][W] instead of pixels: only the patch-widening small-channel kernel reads them (it interleaves while staging)
        if (p->in_planar > 4 || p->in_c != 4 || p->in_w < 4 || p->stride_h < 1 || p->stride_w < 1 || p->add || p->nseg > 1 ||
            !mhip_conv_i8_small_c(p->in_c, p->kw, p->out_c))
            return -2;
        return conv_i8_try_smallc(p, k64); // -2: its patch / LDS budget does not take the shape
    }
    if (mhip_conv_i8_small_c(p->in_c, p->kw, p->out_c)) {
        if (p->stride_h >= 1 && p->stride_w >= 1) {
            int rc = conv_i8_try_rgb(p, k64);
            if (rc == -2) rc = conv_i8_try_smallc(p, k64);
            if (rc != -2) return rc;
        }
        // a patch larger than the small-channel kernel stages (large strides / kernels): the gather kernel reads the same packing
        if (p->add || p->nseg > 1) return -1;
        if (oc_pad % 64 == 0) return launch_generic<64>(p, total_pix, k64);
        return launch_generic<32>(p, total_pix, k64);
    }
    if (p->pre_w) { // fused bottleneck: the patch-staged kernel at the tallest tile that fits (4 rows for few workgroups)
        if (!mhip_zero_page() || !mhip_conv_i8_pre_ok(p)) return -1;
        int th = conv_i8_pre_tile_rows(p);
        const int code = p->variant ? p->variant : tune().variant;
        if (code >= 9 && code <= 11) { // forced (autotuner / tests): that height if it fits
            const int want = code == 10 ? 16 : (code == 9 ? 8 : 4);
            if (conv_i8_patch_ok(p, want, nullptr)) th = want;
        } else if (tune().small_batch && conv_i8_patch_ok(p, 4, nullptr) &&
                   ((long)p->frames * p->out_h * p->out_w + 255) / 256 * ((p->oc_pad + 127) / 128) < 256) {
            th = 4;
        }
        return conv_i8_launch_patch(p, k64, th);
    }
    if ((p->in_c % 16) == 0) {
        if (!mhip_zero_page()) return -1;
        const int nks = k64 / BK;
        if (p->variant < 0 || p->variant > NVARIANTS) return -1;
        int code = p->variant;
        if (p->nseg > 1) {
            if (!seg_valid(p) || !persist_eligible(p)) return -1;
            if (!code && tune().variant) { // forced from outside, as below
                int codes[NVARIANTS];
                const int n = mhip_conv_i8_variants(p, codes, NVARIANTS);
                for (int i = 0; i < n; i++)
                    if (codes[i] == tune().variant) code = codes[i];
            }
            variant_t v = variant_of(code == 4 || code == 14 || code == 15 ? code : 2);
            if (!code) {
                const variant_t dv = default_variant(p, nks);
                v.bpx = dv.bpx ? dv.bpx : 128;
                v.wres = dv.persist && dv.wres;
            }
            return launch_variant(p, total_pix, k64, v);
        }
        if (!code && tune().variant) { // forced from outside: only where this layer has that variant
            int codes[NVARIANTS];
            const int n = mhip_conv_i8_variants(p, codes, NVARIANTS);
            for (int i = 0; i < n; i++)
                if (codes[i] == tune().variant) code = codes[i];
        }
        return launch_variant(p, total_pix, k64, code ? variant_of(code) : default_variant(p, nks));
    }
    if (oc_pad % 64 == 0) return launch_generic<64>(p, total_pix, k64);
    return launch_generic<32>(p, total_pix, k64);
}

