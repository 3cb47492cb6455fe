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
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "../mhip.h"

extern "C" hipStream_t mhip_stream_native(void);
extern "C" int mhip_check(hipError_t e, const char *what);
extern "C" const void *mhip_zero_page(void);

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

#define BP 128      // pixels per workgroup
#define BK 64       // K bytes per step = one MFMA
#define NTHREADS 256
#define OPAD 4      // padding of an epilogue LDS row (bytes): spreads pixel rows over banks
#define LUTB 512    // bytes reserved at LDS address 0 for the fused LUT (256-entry, or the 512-entry half-step form)

// LDS tile row = 64 bytes (4 chunks of 16).  XOR the chunk index with
// ((row>>2)&1)<<1: conflict-free for the ds_read_b128 lane groups of gfx950
// (MI355X_MICROARCH.md, LDS table) when 16 lanes read 16 consecutive rows.
__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * BK + (((chunk ^ ((row >> 1) & 2))) << 4);
}

// One output value: 6 VALU when SAFE.  `lo` is the lower clamp (-128, or 0 for the fused ReLU).
// The +/-0.5 is copysign(0.5, scaled): same result as the reference's `scaled >= 0 ? 0.5f : -0.5f`
// for every input (for -0.0 both roundings truncate to 0; NaN stays NaN).
// SAFE (decided on the host): |acc*cs| can never reach 2^31 and cs is finite, so the x86
// "integer indefinite" fix-up (out of range / NaN -> INT_MIN -> -128) is provably dead.
template <bool SAFE>
__device__ __forceinline__ int requant(int acc, float cs, int lo) {
    const float scaled = (float)acc * cs;
    const float half = __int_as_float((__float_as_int(scaled) & (int)0x80000000) | 0x3f000000);
    const float biased = scaled + half;
    int r = (int)biased;                                    // v_cvt_i32_f32: saturates, NaN -> 0
    if (!SAFE) r = biased < 2147483648.0f ? r : INT_MIN;    // x86 cvttss2si: +overflow and NaN -> INT_MIN
    r = r < lo ? lo : r;
    r = r > 127 ? 127 : r;
    return r;
}

// ---- the epilogue's value path, trimmed to the VALU floor.  The epilogue is VALU-issue bound (a wave64 VALU
// instruction holds its SIMD for 4 cycles, and every output byte of the network passes through here), so each
// instruction per value counts:
//   v_cvt_f32_i32, v_mul_f32, v_bfi_b32 (copysign 0.5), v_add_f32, v_cvt_i32_f32, v_med3_i32   = 6 per value
//   + 3 instructions per 4 values to pack bytes into a dword; the fused LUT costs no VALU at all: ds_read_u8 takes
//   the clamped value (negative included: the LDS address is vaddr + offset modulo 2^32) with offset = LUT + 128.
// LUT0: the 256-byte LUT sits at LDS byte address 0 (first in dynamic LDS of a kernel that owns no static LDS).
__device__ __forceinline__ int requant_safe(int acc, float cs, int lo, int hi) {
    const float scaled = (float)acc * cs;
    const float half = __int_as_float((__float_as_int(scaled) & (int)0x80000000) | 0x3f000000);
    const int r = (int)(scaled + half);
    int m;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(m) : "v"(r), "v"(lo), "v"(hi)); // lo <= hi is not provable for the compiler
    return m;
}
__device__ __forceinline__ uint32_t pack4(int q0, int q1, int q2, int q3) { // low bytes of four ints
    const uint32_t a = __builtin_amdgcn_perm((uint32_t)q1, (uint32_t)q0, 0x0c0c0400u);
    const uint32_t b = __builtin_amdgcn_perm((uint32_t)q3, (uint32_t)q2, 0x0c0c0400u);
    return (b << 16) | a;
}
__device__ __forceinline__ void lut4_at0(int q0, int q1, int q2, int q3, int &v0, int &v1, int &v2, int &v3) {
    asm volatile("ds_read_i8 %0, %4 offset:128\n\tds_read_i8 %1, %5 offset:128\n\t" // sign-extending byte loads
                 "ds_read_i8 %2, %6 offset:128\n\tds_read_i8 %3, %7 offset:128"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                 : "v"(q0), "v"(q1), "v"(q2), "v"(q3)
                 : "memory");
}
__device__ __forceinline__ void lds_base_must_be_zero(const void *dynamic_lds) {
    // every convolution wave runs at raised issue priority: the detection tail of the previous batch shares the SIMDs
    // (its serial sort wave otherwise takes issue slots from a wave its whole workgroup then waits for at the barrier)
    __builtin_amdgcn_s_setprio(3);
    unsigned a = (unsigned)(size_t)(const __attribute__((address_space(3))) void *)dynamic_lds;
    asm volatile("" : "+s"(a)); // opaque: the optimiser assumes a global's address is never 0 and would fold the test
    if (a != 0u) __builtin_trap();
}
// NV = 8 or 16 accumulators of one pixel (consecutive channels) -> NV/4 packed dwords
// fused residual Add (reference mars_runtime.c:835-905, the ADD branch): out = sat8(trunc((v*s_conv + x*s_other)*inv + 0.5f))
// with v the convolution's (LUT-mapped) int8 result and x the other operand's byte; the host fuses only when the
// float -> int conversion is provably in range, so the clamp is a med3.
struct add_args_t {
    float s_conv, s_other, inv;
};
__device__ __forceinline__ int add_one(int v, uint32_t xword, int k, const add_args_t &g, int lo8, int hi8) {
    const int x = __builtin_amdgcn_sbfe((int)xword, 8 * k, 8);
    const float y = (float)v * g.s_conv + (float)x * g.s_other;
    const float t = y * g.inv;
    const int r = (int)(t + 0.5f);
    int m;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(m) : "v"(r), "v"(lo8), "v"(hi8));
    return m;
}
// FAST (host: p.lut2): the fused LUT in its half-step form.  round-half-away(x) = f(trunc(2x)) for every float except
// +-0x3EFFFFFF (checked over all floats below 1000; that value rounds up inside the reference's float add), and the
// host verifies no accumulator of the layer can produce it.  So requantise + clamp + LUT become: v_cvt_f32_i32,
// v_mul_f32 (by 2*cs, exact doubling), v_cvt_i32_f32, v_med3_i32 to [-256, 255], ds_read_i8 from the 512-entry table
// lut2[k + 256] = lut[clamp(f(k), lo, 127) + 128]: 4 instead of 6 VALU per value, the lower clamp folded into the table.
__device__ __forceinline__ void lut4_fast(int q0, int q1, int q2, int q3, int &v0, int &v1, int &v2, int &v3) {
    asm volatile("ds_read_i8 %0, %4 offset:256\n\tds_read_i8 %1, %5 offset:256\n\t"
                 "ds_read_i8 %2, %6 offset:256\n\tds_read_i8 %3, %7 offset:256"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                 : "v"(q0), "v"(q1), "v"(q2), "v"(q3)
                 : "memory");
}
// the same half-step requantisation against a SECOND table at LDS bytes 512..1023 (the 1x1 stage of the fused bottleneck)
template <int NV>
__device__ __forceinline__ void requant_pack_pre(const int (&a)[NV], float cs, uint32_t (&pk)[NV / 4]) {
    const float cs2 = cs * 2.0f;
    const int klo = -256, khi = 255;
    int q[NV], v[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int k = (int)((float)a[i] * cs2);
        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(q[i]) : "v"(k), "v"(klo), "v"(khi));
    }
#pragma unroll
    for (int g = 0; g < NV / 4; g++)
        asm volatile("ds_read_i8 %0, %4 offset:768\n\tds_read_i8 %1, %5 offset:768\n\t"
                     "ds_read_i8 %2, %6 offset:768\n\tds_read_i8 %3, %7 offset:768"
                     : "=&v"(v[4 * g]), "=&v"(v[4 * g + 1]), "=&v"(v[4 * g + 2]), "=&v"(v[4 * g + 3])
                     : "v"(q[4 * g]), "v"(q[4 * g + 1]), "v"(q[4 * g + 2]), "v"(q[4 * g + 3])
                     : "memory");
    if (NV == 16)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                       "+v"(v[8 % NV]), "+v"(v[9 % NV]), "+v"(v[10 % NV]), "+v"(v[11 % NV]), "+v"(v[12 % NV]), "+v"(v[13 % NV]),
                       "+v"(v[14 % NV]), "+v"(v[15 % NV]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
    for (int g = 0; g < NV / 4; g++) pk[g] = pack4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}
template <int NV, bool HAS_LUT, bool SAFE, bool LUT0, bool ADD = false, bool FAST = false>
__device__ __forceinline__ void requant_pack(const int (&a)[NV], float cs, int lo, const uint8_t *lut128, uint32_t (&pk)[NV / 4],
                                             const uint32_t *xw = nullptr, const add_args_t *ga = nullptr) {
    int q[NV];
    const int hi = 127;
    if (FAST && HAS_LUT && LUT0 && SAFE) {
        const float cs2 = cs * 2.0f;
        const int klo = -256, khi = 255;
        int v[NV];
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int k = (int)((float)a[i] * cs2);
            asm("v_med3_i32 %0, %1, %2, %3" : "=v"(q[i]) : "v"(k), "v"(klo), "v"(khi));
        }
#pragma unroll
        for (int g = 0; g < NV / 4; g++)
            lut4_fast(q[4 * g], q[4 * g + 1], q[4 * g + 2], q[4 * g + 3], v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
        if (NV == 16)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                           "+v"(v[8 % NV]), "+v"(v[9 % NV]), "+v"(v[10 % NV]), "+v"(v[11 % NV]), "+v"(v[12 % NV]), "+v"(v[13 % NV]),
                           "+v"(v[14 % NV]), "+v"(v[15 % NV]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
        if (ADD) {
            const int lo8 = -128;
#pragma unroll
            for (int i = 0; i < NV; i++) v[i] = add_one(v[i], xw[i >> 2], i & 3, *ga, lo8, hi);
        }
#pragma unroll
        for (int g = 0; g < NV / 4; g++) pk[g] = pack4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
        return;
    }
#pragma unroll
    for (int i = 0; i < NV; i++) q[i] = SAFE ? requant_safe(a[i], cs, lo, hi) : requant<false>(a[i], cs, lo);
    if (HAS_LUT && LUT0) {
        int v[NV];
#pragma unroll
        for (int g = 0; g < NV / 4; g++)
            lut4_at0(q[4 * g], q[4 * g + 1], q[4 * g + 2], q[4 * g + 3], v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
        // the compiler does not count LDS loads issued from asm: wait here, and thread the values through the wait
        if (NV == 16)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                           "+v"(v[8 % NV]), "+v"(v[9 % NV]), "+v"(v[10 % NV]), "+v"(v[11 % NV]), "+v"(v[12 % NV]), "+v"(v[13 % NV]),
                           "+v"(v[14 % NV]), "+v"(v[15 % NV]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
        if (ADD) {
            const int lo8 = -128;
#pragma unroll
            for (int i = 0; i < NV; i++) v[i] = add_one(v[i], xw[i >> 2], i & 3, *ga, lo8, hi);
        }
#pragma unroll
        for (int g = 0; g < NV / 4; g++) pk[g] = pack4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
    } else {
        if (HAS_LUT) {
#pragma unroll
            for (int i = 0; i < NV; i++) q[i] = (int8_t)lut128[q[i]];
        }
        if (ADD) {
            const int lo8 = -128;
#pragma unroll
            for (int i = 0; i < NV; i++) q[i] = add_one(q[i], xw[i >> 2], i & 3, *ga, lo8, hi);
        }
#pragma unroll
        for (int g = 0; g < NV / 4; g++) pk[g] = pack4(q[4 * g], q[4 * g + 1], q[4 * g + 2], q[4 * g + 3]);
    }
}

// exact unsigned division by a launch-time constant (Granlund & Montgomery, N = 32):
// q = mulhi(m, n); q = (q + ((n - q) >> s1)) >> s2
struct fastdiv_t {
    unsigned m, s1, s2;
};
__device__ __forceinline__ unsigned fdiv(unsigned n, const fastdiv_t d) {
    const unsigned q = __umulhi(d.m, n);
    return (q + ((n - q) >> d.s1)) >> d.s2;
}
static fastdiv_t make_fastdiv(unsigned d) {
    fastdiv_t r;
    unsigned l = 0;
    while ((1ull << l) < d) l++;
    r.m = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    r.s1 = l < 1 ? l : 1;
    r.s2 = l > 0 ? l - 1 : 0;
    return r;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void glds16(const void *gsrc, void *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// LDS-DMA through a buffer resource: a lane whose offset is out of range delivers ZEROS to LDS (probed), so taps
// outside the image need no zero page and no 64-bit address select; offsets are 32-bit.
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rs, int voffset, int soffset, void *lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)lds_wave_base, 16, voffset, soffset, 0, 0);
}

// XCD-aware block order: the 8 XCDs take consecutive dispatch ids round-robin; give each
// XCD one contiguous range of logical tiles so tiles that share input rows / weight
// panels share an L2 (bijective for any grid size).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned q = nblk >> 3, r = nblk & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// ---------------------------------------------------------------------------------
// shared epilogue: accumulators -> requant -> LDS tile -> coalesced global stores
// `rowoff[row]` = byte offset of tile row `row`'s pixel in the output (frame*out_stride + rem*out_c for NHWC,
// frame*out_stride + rem for NCHW), or -1 when the row is outside the image/batch; filled by fill_rowoff().
template <int BPX, class PixelOf>
__device__ __forceinline__ void fill_rowoff(const mhip_conv_i8_t &p, long *rowoff, PixelOf pixel_of, unsigned hw,
                                            const fastdiv_t dhw) {
    if (threadIdx.x < BPX) {
        const long pix = pixel_of((int)threadIdx.x); // global pixel index (frame*H*W + y*W + x) or -1
        long off = -1;
        if (pix >= 0) {
            const unsigned f = fdiv((unsigned)pix, dhw), rem = (unsigned)pix - f * hw;
            off = (long)f * (long)p.out_stride +
                  (p.out_nchw ? (long)rem : (long)rem * (p.out_pix_stride ? p.out_pix_stride : p.out_c) + p.out_ch_off);
        }
        rowoff[threadIdx.x] = off;
    }
}

// Output channels are PERMUTED inside each wave's channel range (host packer, mhip_conv_i8_oc_row): MFMA row
// (lane>>4)*4 + r of oc-subtile s carries channel (lane>>4)*4*WOC + s*4 + r, so the WOC*4 results a lane holds
// for one pixel are CONSECUTIVE channels.
//  DIRECT (NHWC, 16-byte aligned rows): one 16-byte (WOC=4) / 8-byte (WOC=2) global store per pixel straight
//  from registers -- no LDS tile, no barrier.  Otherwise the int8 tile is staged in LDS and copied out coalesced.
template <int BPX, int BN, int WPX, int WOC, bool HAS_LUT, bool SAFE, bool DIRECT, bool LUT0>
__device__ __forceinline__ void epilogue_t(const mhip_conv_i8_t &p, v4i (&acc)[WOC][WPX], int8_t *tile, const uint8_t *slut,
                                           const long *rowoff, int oc0, int pxw, int ocw, int hw) {
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int ROW = BN + OPAD;
    const int lo = p.relu ? 0 : -128; // fused ReLU == raising the lower clamp
    const uint8_t *lut128 = slut + 128;
    const int chan = ocw + (lane >> 4) * (4 * WOC); // first of this lane's WOC*4 consecutive channels (tile-relative)
#pragma unroll
    for (int t = 0; t < WPX; t++) {
        const int prow = pxw + t * 16 + (lane & 15);
        uint32_t pk[WOC];
        int a[WOC * 4]; // the bias is already inside the accumulators
#pragma unroll
        for (int s = 0; s < WOC; s++)
#pragma unroll
            for (int r = 0; r < 4; r++) a[s * 4 + r] = acc[s][t][r];
        if (DIRECT && SAFE && p.add) { // fused residual Add: the other operand has the output's layout
            const long off = rowoff[prow];
            const bool ok = off >= 0 && oc0 + chan < p.out_c;
            uint32_t xw[WOC];
#pragma unroll
            for (int s = 0; s < WOC; s++) xw[s] = 0;
            if (ok) {
                const int8_t *x = p.add + off + oc0 + chan;
                if (WOC == 4) { const v4i t4 = *(const v4i *)x; xw[0] = t4[0]; xw[1] = t4[1]; xw[WOC > 2 ? 2 : 0] = t4[2]; xw[WOC > 3 ? 3 : 0] = t4[3]; }
                else { const uint2 t2 = *(const uint2 *)x; xw[0] = t2.x; xw[WOC > 1 ? 1 : 0] = t2.y; }
            }
            const add_args_t ga = {p.add_s_conv, p.add_s_other, p.add_inv};
            if (HAS_LUT && LUT0 && p.lut2) requant_pack<WOC * 4, HAS_LUT, true, LUT0, true, true>(a, p.cs, lo, lut128, pk, xw, &ga);
            else requant_pack<WOC * 4, HAS_LUT, true, LUT0, true>(a, p.cs, lo, lut128, pk, xw, &ga);
        } else {
            if (HAS_LUT && LUT0 && SAFE && p.lut2) requant_pack<WOC * 4, HAS_LUT, SAFE, LUT0, false, true>(a, p.cs, lo, lut128, pk);
            else requant_pack<WOC * 4, HAS_LUT, SAFE, LUT0>(a, p.cs, lo, lut128, pk);
        }
        if (DIRECT) {
            const long off = rowoff[prow];
            if (off >= 0 && oc0 + chan < p.out_c) {
                int8_t *d = p.out + off + oc0 + chan;
                if (WOC == 4) *(v4i *)d = (v4i){(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]};
                else *(uint2 *)d = make_uint2(pk[0], pk[WOC > 1 ? 1 : 0]);
            }
        } else {
#pragma unroll
            for (int s = 0; s < WOC; s++) *(uint32_t *)(tile + prow * ROW + chan + s * 4) = pk[s];
        }
    }
    if (DIRECT) return;
    __syncthreads();
    const int ncols = p.out_c - oc0 < BN ? p.out_c - oc0 : BN; // valid channels of this tile
    if (!p.out_nchw && ((p.out_c | p.out_pix_stride | p.out_ch_off) & 15) == 0) {
        constexpr int CPR = BN / 16; // 16-byte chunks per pixel row
        for (int id = tid; id < BPX * CPR; id += (int)blockDim.x) {
            const int row = id / CPR, c = id - row * CPR;
            const long off = rowoff[row];
            if (off < 0 || c * 16 >= ncols) continue;
            const int8_t *s = tile + row * ROW + c * 16;
            v4i v = {*(const int *)s, *(const int *)(s + 4), *(const int *)(s + 8), *(const int *)(s + 12)};
            *(v4i *)(p.out + off + oc0 + c * 16) = v;
        }
    } else if (!p.out_nchw) { // e.g. the 255-channel heads: rows are not 16-byte aligned in HBM
        constexpr int CPR = BN / 16;
        for (int id = tid; id < BPX * CPR; id += (int)blockDim.x) {
            const int row = id / CPR, c = id - row * CPR;
            const long off = rowoff[row];
            if (off < 0 || c * 16 >= ncols) continue;
            const int8_t *s = tile + row * ROW + c * 16;
            int8_t *d = p.out + off + oc0 + c * 16;
            if (c * 16 + 16 <= ncols) { // unaligned dwordx4 store (gfx950 accepts any byte alignment)
                v4i v = {*(const int *)s, *(const int *)(s + 4), *(const int *)(s + 8), *(const int *)(s + 12)};
                __builtin_memcpy(d, &v, 16);
            } else { // ragged end of the pixel row (255 channels: 15 bytes): 8 + 4 + 2 + 1, any alignment
                const int rem = ncols - c * 16;
                int o = 0;
                if (rem & 8) { __builtin_memcpy(d, s, 8); o = 8; }
                if (rem & 4) { __builtin_memcpy(d + o, s + o, 4); o += 4; }
                if (rem & 2) { __builtin_memcpy(d + o, s + o, 2); o += 2; }
                if (rem & 1) d[o] = s[o];
            }
        }
    } else { // [O][H][W]: consecutive lanes -> consecutive pixels of one channel
        for (int id = tid; id < BPX * BN; id += (int)blockDim.x) {
            const int c = id / BPX, row = id - c * BPX;
            const long off = rowoff[row];
            if (off < 0 || c >= ncols) continue;
            p.out[off + (size_t)(oc0 + c) * hw] = tile[row * ROW + c];
        }
    }
}

template <int BPX, int BN, int WPX, int WOC, bool LUT0 = false>
__device__ __forceinline__ void epilogue(const mhip_conv_i8_t &p, v4i (&acc)[WOC][WPX], int8_t *tile, const uint8_t *slut,
                                         const long *rowoff, int oc0, int pxw, int ocw, int hw) {
    const bool direct = !p.out_nchw && ((p.out_c | p.out_pix_stride | p.out_ch_off) & 15) == 0;
#define EPI(L, S, D) epilogue_t<BPX, BN, WPX, WOC, L, S, D, LUT0>(p, acc, tile, slut, rowoff, oc0, pxw, ocw, hw)
    if (direct) {
        if (p.lut) { if (p.safe) EPI(true, true, true); else EPI(true, false, true); }
        else { if (p.safe) EPI(false, true, true); else EPI(false, false, true); }
    } else {
        if (p.lut) { if (p.safe) EPI(true, true, false); else EPI(true, false, false); }
        else { if (p.safe) EPI(false, true, false); else EPI(false, false, false); }
    }
#undef EPI
}

// accumulators start at the bias: lane holds channels ocbase + s*16 + (lane>>4)*4 .. +3 of every pixel subtile
template <int WPX, int WOC>
__device__ __forceinline__ void init_acc(const mhip_conv_i8_t &p, v4i (&acc)[WOC][WPX], int ocbase) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int s = 0; s < WOC; s++) {
        const v4i b = p.bias ? *(const v4i *)(p.bias + ocbase + s * 16 + (lane >> 4) * 4) : (v4i){0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < WPX; t++) acc[s][t] = b;
    }
}

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
template <int BPX, int BN, int STAGES, int KS = 1, int NW = 4>
__global__ __launch_bounds__(NW * 64) void conv_i8_mfma(const mhip_conv_i8_t p, const long total_pix, const int k64,
                                                         const int8_t *__restrict__ zeros, const unsigned noc,
                                                         const unsigned nblk, const int lg_inc, const unsigned kw_magic,
                                                         const fastdiv_t dhw, const fastdiv_t dow, const int bufmode,
                                                         const unsigned in_bytes) {
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
                blds16(xrs, ok ? xvoff[j] + ukoff : -1, 0, sb + (wv * XROWS + j * 16) * BK);
            }
#pragma unroll
            for (int j = 0; j < LW; j++) blds16(wrs, wvoff[j], ks * BK, sb + BPX * BK + wq[j] * 16 * BK);
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

    const int nst = nks / KS; // ring stages to run (KS == 2: the host guarantees an even nks)
#pragma unroll
    for (int s = 0; s < STAGES - 1; s++)
        if (s < nst)
#pragma unroll
            for (int u = 0; u < KS; u++) issue(s * KS + u, s);

    const int frow = lane & 15, fchunk = lane >> 4;
    int stage = 0, nstage = STAGES - 1;
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
                for (int t = 0; t < WPX; t++) acc[s][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[s], xb[t], acc[s][t], 0, 0, 0);
        }
        stage = stage + 1 == STAGES ? 0 : stage + 1;
        nstage = nstage + 1 == STAGES ? 0 : nstage + 1;
    }
    __syncthreads(); // every wave is done reading the ring: reuse it for the output tile
    epilogue<BPX, BN, WPX, WOC, true>(p, acc, lds, slut, rowoff, oc0, pxw, ocw, hw);
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

    v4i bias[WOC];
#pragma unroll
    for (int s = 0; s < WOC; s++)
        bias[s] = p.bias ? *(const v4i *)(p.bias + oc0 + ocw + s * 16 + (lane >> 4) * 4) : (v4i){0, 0, 0, 0};
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
        wq[j] = BN >= 64 ? wv * LW + j : (wv & 1);
        wsrc[j] = p.w + (size_t)(oc0 + wq[j] * 16 + (lane >> 2)) * k64 + schunk * 16;
        wvoff[j] = (oc0 + wq[j] * 16 + (lane >> 2)) * k64 + schunk * 16;
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
            else
                __builtin_amdgcn_raw_buffer_store_b64((v2i){(int)pk[0], (int)pk[WOC > 1 ? 1 : 0]}, orsrc, voff, 0, 0);
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
}

// ---------------------------------------------------------------------------------
// conv_i8_r128p (launch variant 20): the 128-byte-K-step tile as ONE persistent 8-wave workgroup per CU with a THREE-stage
// ring (3 x (256 + 128) x 128 B = 144 KB of LDS) that runs across tile boundaries.  The argument (DESIGN.md section 5,
// "Deep-K"): the L2 -> LDS DMA delivers ~27 B/clk/CU in 64-byte row pieces and ~40 in whole 128-byte lines, but only with
// enough bytes in flight (latency x bandwidth ~ 96 KB per CU); conv_i8_r128 has the lines but one stage in flight, the
// 8-wave 64-byte tile has 96 KB in flight but the pieces.  Here two 48 KB stages are in flight behind the one being
// multiplied, and the K stream of tile i+1 is already arriving while tile i is requantised (stores go out as raw buffer
// stores, always the same number per wave, so vmcnt is counted across them as in conv_i8_persist).
template <bool HAS_LUT>
__global__ __launch_bounds__(512) void conv_i8_r128p(const mhip_conv_i8_t p, const unsigned total_pix, const int k128,
                                                      const unsigned noc, const unsigned npt, const unsigned ngrp, const int lg_inc,
                                                      const unsigned kw_magic, const fastdiv_t dhw, const fastdiv_t dow,
                                                      const unsigned in_bytes, const unsigned out_bytes) {
    constexpr int BPX = 256, BN = 128, NW = 8, STG = 3;
    constexpr int STAGE = (BPX + BN) * 128;
    constexpr int NWN = 2, NWM = NW / NWN;
    constexpr int WPX = BPX / NWM / 16; // 4
    constexpr int WOC = BN / NWN / 16;  // 4
    constexpr int XI = BPX / 8 / NW;    // 4 DMA instructions of 8 rows x 128 B for the pixel tile
    constexpr int WI = BN / 8 / NW;     // 2 for the weight tile
    constexpr int L = XI + WI;          // per wave and stage
    constexpr int NST = WPX;            // buffer stores per wave and tile
    extern __shared__ __attribute__((aligned(16))) int8_t dynlds[];
    uint8_t *slut = (uint8_t *)dynlds; // LDS byte address 0 (requant_pack LUT0)
    int8_t *lds = dynlds + LUTB;
    lds_base_must_be_zero(dynlds);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned id = xcd_remap(blockIdx.x, noc * ngrp);
    const unsigned grp = id / noc, ot = id - grp * noc;
    const int oc0 = (int)ot * BN;
    const unsigned t0 = (unsigned)(((unsigned long long)grp * npt) / ngrp);
    const unsigned t1 = (unsigned)(((unsigned long long)(grp + 1) * npt) / ngrp);
    if (t0 >= t1) return;
    const unsigned hw = (unsigned)(p.out_h * p.out_w);
    const int wm = wv % NWM, wn = wv / NWM;
    const int pxw = wm * (WPX * 16), ocw = wn * (WOC * 16);

    v4i bias[WOC];
#pragma unroll
    for (int q = 0; q < WOC; q++)
        bias[q] = p.bias ? *(const v4i *)(p.bias + oc0 + ocw + q * 16 + (lane >> 4) * 4) : (v4i){0, 0, 0, 0};
    if (HAS_LUT) {
        if (p.lut2) { if (tid < 128) ((uint32_t *)slut)[tid] = ((const uint32_t *)p.lut2)[tid]; }
        else if (tid < 64) ((uint32_t *)slut)[tid] = ((const uint32_t *)p.lut)[tid];
        __syncthreads();
    }
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, (int)in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, p.oc_pad * k128, 0x00020000);

    // DMA assignment: instruction j of this wave covers tile rows (wv * XI + j) * 8 .. + 7; lane i -> row i / 8, slot i % 8
    const int lrow = lane >> 3, lslot = lane & 7;
    int xvoff[XI];
    unsigned tapmask[XI];
    auto setup_rows = [&](unsigned tile) {
#pragma unroll
        for (int j = 0; j < XI; j++) {
            const int trow = (wv * XI + j) * 8 + lrow;
            const unsigned pix = tile * BPX + (unsigned)trow;
            const bool valid = pix < total_pix;
            const unsigned f = valid ? fdiv(pix, dhw) : 0u;
            const unsigned rem = valid ? pix - f * hw : 0u;
            const int oy = (int)fdiv(rem, dow), ox = (int)(rem - (unsigned)oy * (unsigned)p.out_w);
            const int iy0 = oy * p.stride_h - p.pad_top, ix0 = ox * p.stride_w - p.pad_left;
            const int chunk = lslot ^ ((trow >> 1) & 7);
            xvoff[j] = (int)(f * (unsigned)p.in_stride) + (iy0 * p.in_w + ix0) * p.in_c + chunk * 16; // only used with in-image taps
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
    int wvoff[WI];
#pragma unroll
    for (int j = 0; j < WI; j++) {
        const int trow = (wv * WI + j) * 8 + lrow;
        wvoff[j] = (oc0 + trow) * k128 + (lslot ^ ((trow >> 1) & 7)) * 16;
    }
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

    // issue cursor over the (tile, stage) stream; younger[i] = vector-memory instructions this wave has issued after the
    // i-th oldest stage still in the ring
    unsigned itile = t0;
    int ist = 0;
    int younger[STG - 1];
    setup_rows(t0);
    int nbuf = 0;
#pragma unroll
    for (int s2 = 0; s2 < STG - 1; s2++) {
        int n = 0;
        if (itile < t1) {
            issue(ist, nbuf);
            n = L;
            if (++ist == nst) {
                ist = 0;
                if (++itile < t1) setup_rows(itile);
            }
        }
#pragma unroll
        for (int i = 0; i < s2; i++) younger[i] += n;
        younger[s2] = 0;
        nbuf++;
    }
    nbuf = STG - 1;

    const int frow = lane & 15, fchunk = lane >> 4;
    const int chan = ocw + (lane >> 4) * (4 * WOC);
    const int pstride = p.out_pix_stride ? p.out_pix_stride : p.out_c;
    const int lo = p.relu ? 0 : -128;
    const uint8_t *lut128 = slut + 128;
    int buf = 0;
    for (unsigned tile = t0; tile < t1; tile++) {
        v4i acc[WOC][WPX];
        for (int st = 0; st < nst; st++) {
            wait_vmcnt_at_most<(STG - 2) * L, NST>(younger[0]);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int8_t *xs = lds + buf * STAGE, *ws = xs + BPX * 128;
            v4i xb[2][WPX], wa[2][WOC];
#pragma unroll
            for (int h = 0; h < 2; h++) {
#pragma unroll
                for (int t = 0; t < WPX; t++) xb[h][t] = *(const v4i *)(xs + lds_off128(pxw + t * 16 + frow, h * 4 + fchunk));
#pragma unroll
                for (int q = 0; q < WOC; q++) wa[h][q] = *(const v4i *)(ws + lds_off128(ocw + q * 16 + frow, h * 4 + fchunk));
            }
            int n = 0;
            if (itile < t1) { // into the buffer multiplied one step ago: every wave is past the barrier above, hence past its reads
                issue(ist, nbuf);
                n = L;
                if (++ist == nst) {
                    ist = 0;
                    if (++itile < t1) setup_rows(itile);
                }
            }
#pragma unroll
            for (int i = 0; i + 1 < STG - 1; i++) younger[i] = younger[i + 1] + n;
            younger[STG - 2] = 0;
            if (st == 0) {
#pragma unroll
                for (int q = 0; q < WOC; q++)
#pragma unroll
                    for (int t = 0; t < WPX; t++) acc[q][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[0][q], xb[0][t], bias[q], 0, 0, 0);
            } else {
#pragma unroll
                for (int q = 0; q < WOC; q++)
#pragma unroll
                    for (int t = 0; t < WPX; t++) acc[q][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[0][q], xb[0][t], acc[q][t], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < WOC; q++)
#pragma unroll
                for (int t = 0; t < WPX; t++) acc[q][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[1][q], xb[1][t], acc[q][t], 0, 0, 0);
            buf = buf + 1 == STG ? 0 : buf + 1;
            nbuf = nbuf + 1 == STG ? 0 : nbuf + 1;
        }
        // epilogue: requantise, optional LUT, one 16-byte buffer store per pixel subtile
#pragma unroll
        for (int t = 0; t < WPX; t++) {
            const unsigned pix = tile * BPX + pxw + t * 16 + (lane & 15);
            const unsigned f = fdiv(pix, dhw), rem = pix - f * hw;
            const unsigned off = f * (unsigned)p.out_stride + rem * (unsigned)pstride + (unsigned)(p.out_ch_off + oc0 + chan);
            const bool ok = pix < total_pix && oc0 + chan + WOC * 4 <= p.out_c;
            uint32_t pk[WOC];
            int a[WOC * 4];
#pragma unroll
            for (int q = 0; q < WOC; q++)
#pragma unroll
                for (int r = 0; r < 4; r++) a[q * 4 + r] = acc[q][t][r];
            if (HAS_LUT && p.lut2) requant_pack<WOC * 4, HAS_LUT, true, true, false, true>(a, p.cs, lo, lut128, pk);
            else requant_pack<WOC * 4, HAS_LUT, true, true>(a, p.cs, lo, lut128, pk);
            __builtin_amdgcn_raw_buffer_store_b128((v4i){(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]}, orsrc, ok ? (int)off : -1, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < STG - 1; i++) younger[i] += NST;
    }
}

// ---------------------------------------------------------------------------------
// patch-staged kernel: k x k convolutions on wide feature maps with few channels (the 160x160 / 80x80 layers of
// yolov5: in_c 32..128).  The implicit-GEMM kernels above fetch every input pixel once per kernel tap through the
// 64 B/clk L1 path, which is what bounds these layers (few output channels per fetched byte).  Here a workgroup
// stages the input patch of a TH x 16 output tile ONCE in LDS (LDS-DMA, double buffered, next tile's patch in
// flight while this one is computed), keeps the weights of its channel tile resident in LDS for its whole
// (persistent) life, and feeds the MFMAs of all taps from LDS: HBM/L2 bytes are read once, the K loop has no
// barrier and no global access at all.
//   B operand of pixel (oy,ox), K chunk (ky,kx,c16) = patch[(oy*s+ky)][(ox*s+kx)][c16]: in NHWC a chunk never
//   straddles pixels, so its LDS address is Ubase(oy,ox) + dU(ky,kx,c16) in 16-byte units, dU tabulated per K step.
//   Stride 2: patch columns are stored de-interleaved (even columns, then odd), so 16 consecutive output pixels
//   read 16 consecutive patch pixels for every tap.  Bank conflicts: 16-byte unit U goes to U ^ ((U>>3) & M),
//   M = 0 / 2 / 6 for in_c = 32 / 64 / 128 -- with it the lane groups of ds_read_b128 touch 16 distinct
//   16-byte bank groups for any tap (derivation: DESIGN.md section 5).
#define PT_TW 16
#define PT_NIMAX 10
// PRE (fused C3 bottleneck, stride 1, in_c 32 / 64): t = SiLU(conv1x1(x)) is evaluated on the staged patch of x -- halo
// included, zero where the pixel lies outside the image (the k x k convolution's SAME padding applies to t) -- into a
// second patch buffer, and the K loop reads that one: t never goes to HBM.  The 1x1's weights ([in_c][64], K padded
// with zeros) and its half-step table (LDS bytes 512..1023) stay resident like the main weights.
template <int TH, int BN, bool HAS_LUT, bool PRE = false>
__global__ __launch_bounds__(NTHREADS) void conv_i8_patch(const mhip_conv_i8_t p, const int k64, const int tiles_x,
                                                          const int tiles_y, const unsigned ntiles_all, const int PH,
                                                          const int PW, const int PWP, const int PWH, const int ni,
                                                          const int8_t *__restrict__ zeros, const fastdiv_t dtx,
                                                          const fastdiv_t dty, const fastdiv_t dpwp, const unsigned out_bytes,
                                                          const int dbl, const int xmap) {
    constexpr int WPX = TH / 4;  // tile rows (16-pixel subtiles) per wave
    constexpr int WOC = BN / 16; // every wave covers all BN channels of its rows
    constexpr int NST = WPX;
    extern __shared__ __attribute__((aligned(16))) int8_t dynlds[];
    uint8_t *slut = (uint8_t *)dynlds; // LDS byte address 0 (requant_pack LUT0)
    lds_base_must_be_zero(dynlds);
    const int nks = k64 / BK;
    constexpr int LB = LUTB + (PRE ? 512 : 0);             // PRE: the 1x1's table behind the main one
    int *dutab = (int *)(dynlds + LB);                     // [nks][4] unit offsets of the K chunks
    int8_t *wl = dynlds + LB + ((nks * 16 + 255) & ~255);   // [nks][BN][64], swizzled like the ring tiles
    const int patch_bytes = ni * 4096;                     // whole DMA instructions (4 waves x 1 KB)
    int8_t *w1l = wl + nks * BN * BK;                      // PRE: [in_c][64] weights of the 1x1
    int8_t *patch0 = w1l + (PRE ? p.in_c * BK : 0);
    int8_t *tpatch = patch0 + (dbl ? 2 : 1) * patch_bytes; // PRE: the patch of t

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int oc0 = blockIdx.y * BN;
    const int C = p.in_c, lgc = 31 - __builtin_clz((unsigned)C), cpp = C >> 4, lgcpp = lgc - 4;
    const int s = p.stride_w;
    const unsigned M = C >= 128 ? 6u : (C >= 64 ? 2u : 0u);

    v4i bias[WOC];
#pragma unroll
    for (int q = 0; q < WOC; q++) bias[q] = p.bias ? *(const v4i *)(p.bias + oc0 + q * 16 + (lane >> 4) * 4) : (v4i){0, 0, 0, 0};
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
    // this lane's units of the patch DMA: instruction n of wave wv fills physical units (n*4+wv)*64 + lane
    int uoff[PT_NIMAX], upos[PT_NIMAX]; // byte offset from the tile's first input pixel; (py << 16) | px, or -1
#pragma unroll
    for (int n = 0; n < PT_NIMAX; n++) {
        uoff[n] = 0;
        upos[n] = -1;
        if (n < ni) {
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
    auto issue_patch = [&](unsigned t, int8_t *dst) {
        int tx, ty;
        unsigned f;
        tile_xy(t, tx, ty, f);
        const int iy0 = ty * TH * s - p.pad_top, ix0 = tx * PT_TW * s - p.pad_left;
        const int8_t *base = p.in + (size_t)f * p.in_stride + ((long)iy0 * p.in_w + ix0) * C;
#pragma unroll
        for (int n = 0; n < PT_NIMAX; n++)
            if (n < ni) {
                const int py = upos[n] >> 16, px = upos[n] & 0xffff;
                const bool ok = upos[n] >= 0 && (unsigned)(iy0 + py) < (unsigned)p.in_h && (unsigned)(ix0 + px) < (unsigned)p.in_w;
                glds16(ok ? base + uoff[n] : zeros, dst + (n * 4 + wv) * 1024);
            }
    };

    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)out_bytes, 0x00020000);
    const int frow = lane & 15, fchunk = lane >> 4;
    const int chan = (lane >> 4) * (4 * WOC);
    const int pstride = p.out_pix_stride ? p.out_pix_stride : p.out_c;
    const int lo = p.relu ? 0 : -128;
    const uint8_t *lut128 = slut + 128;
    int ubase[WPX]; // 16-byte unit of (tile row, column frow), tap (0,0), channel 0
#pragma unroll
    for (int u = 0; u < WPX; u++) ubase[u] = ((wv * WPX + u) * s * PWP + frow) * cpp;

    unsigned t = blockIdx.x;
    int buf = 0;
    if (dbl && t < ntiles) issue_patch(t, patch0);
    bool first = true;
    for (; t < ntiles; t += gridDim.x) {
        if (dbl) {
            // this tile's patch (and, the first time, the weights / tables) has landed; stores of the previous
            // tile are younger than it and may stay in flight
            if (first) wait_vmcnt<0>();
            else wait_vmcnt<NST>();
            __syncthreads();
            const unsigned tn = t + gridDim.x;
            if (tn < ntiles) issue_patch(tn, patch0 + (buf ^ 1) * patch_bytes); // every wave is past its reads of that buffer
        } else {
            // one patch buffer (large stride-2 patches): the co-resident workgroup computes while this one loads
            if (!first) __syncthreads(); // every wave is past its reads of the previous tile
            issue_patch(t, patch0);
            wait_vmcnt<0>();
            __syncthreads();
        }
        first = false;
        const int8_t *patch = patch0 + buf * patch_bytes;
        int tx, ty;
        unsigned f;
        tile_xy(t, tx, ty, f);
        if (PRE) {
            // stage 1: t = SiLU(requant(W1 x + b1)) for every pixel of the patch, 16 flat pixels per MFMA column block;
            // lane (i, g) ends with the 4 * WOC1 consecutive channels g * 4 * WOC1 .. of pixel i (row order of the packer)
            constexpr int W1MAX = 4;
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
                        const v4i b1 = *(const v4i *)(p.pre_bias + q * 16 + (lane >> 4) * 4);
                        const v4i r = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, xb1, b1, 0, 0, 0);
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
                        const v4i b1 = *(const v4i *)(p.pre_bias + q * 16 + (lane >> 4) * 4);
                        const v4i r = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, xb1, b1, 0, 0, 0);
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
            __syncthreads(); // t complete: the K loop below reads it in place of x
            patch = tpatch;
        }
        // output offsets of this lane's pixels; with a fused residual Add the other operand (same layout) is fetched
        // now, ahead of the K loop, so its latency never shows
        int voffs[WPX];
        uint32_t xw[WPX][WOC];
#pragma unroll
        for (int u = 0; u < WPX; u++) {
            const int oy = ty * TH + wv * WPX + u, ox = tx * PT_TW + frow;
            const unsigned off = f * (unsigned)p.out_stride + (unsigned)(oy * p.out_w + ox) * (unsigned)pstride +
                                 (unsigned)(p.out_ch_off + oc0 + chan);
            const bool ok = oy < p.out_h && ox < p.out_w && oc0 + chan < p.out_c;
            voffs[u] = ok ? (int)off : -1;
#pragma unroll
            for (int q = 0; q < WOC; q++) xw[u][q] = 0;
            if (p.add && ok) {
                const int8_t *x = p.add + off;
                if (WOC == 4) { const v4i t4 = *(const v4i *)x; xw[u][0] = t4[0]; xw[u][1] = t4[1]; xw[u][WOC > 2 ? 2 : 0] = t4[2]; xw[u][WOC > 3 ? 3 : 0] = t4[3]; }
                else { const uint2 t2 = *(const uint2 *)x; xw[u][0] = t2.x; xw[u][WOC > 1 ? 1 : 0] = t2.y; }
            }
        }
        v4i acc[WOC][WPX];
        for (int ks = 0; ks < nks; ks++) {
            const int du = dutab[ks * 4 + fchunk];
            v4i xb[WPX];
#pragma unroll
            for (int u = 0; u < WPX; u++) {
                const unsigned U = (unsigned)(ubase[u] + du);
                xb[u] = *(const v4i *)(patch + ((U ^ ((U >> 3) & M)) << 4));
            }
            const int8_t *ws = wl + ks * BN * BK;
            if (ks == 0) {
#pragma unroll
                for (int q = 0; q < WOC; q++) {
                    const v4i wa = *(const v4i *)(ws + lds_off(q * 16 + frow, fchunk));
#pragma unroll
                    for (int u = 0; u < WPX; u++) acc[q][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, xb[u], bias[q], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int q = 0; q < WOC; q++) {
                    const v4i wa = *(const v4i *)(ws + lds_off(q * 16 + frow, fchunk));
#pragma unroll
                    for (int u = 0; u < WPX; u++) acc[q][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, xb[u], acc[q][u], 0, 0, 0);
                }
            }
        }
        if (dbl) buf ^= 1;
#pragma unroll
        for (int u = 0; u < WPX; u++) {
            uint32_t pk[WOC];
            int a[WOC * 4];
#pragma unroll
            for (int q = 0; q < WOC; q++)
#pragma unroll
                for (int r = 0; r < 4; r++) a[q * 4 + r] = acc[q][u][r];
            const bool fast = HAS_LUT && p.lut2 != nullptr;
            if (p.add) {
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
    }
}

// ---------------------------------------------------------------------------------
// patch-staged input, streamed weights: k x k convolutions with 64 / 128 input channels and 128+ output channels (the
// deep 3x3 layers).  The implicit-GEMM forms move (256 + 128) x 64 bytes into LDS per K step and their K loop waits
// for that DMA (20 B/clk/CU arrive, 47 would keep the MFMAs busy).  Here the input patch of a 16 x 16 output tile is
// staged ONCE (as conv_i8_patch does) and serves all k*k taps from LDS; only the 128 x 64 weight bytes of a K step
// stream through a 3-stage ring: 188 instead of 442 KB of LDS-DMA per tile of a 3x3 128 -> 128 layer.  8 waves
// (4 pixel-row groups x 2 channel halves, 64 x 64 accumulators each), one tile per workgroup, two workgroups per CU.
#define PWS_NIMAX 7
template <bool HAS_LUT>
__global__ __launch_bounds__(512) void conv_i8_patchw(const mhip_conv_i8_t p, const int k64, const int tiles_x, const int tiles_y,
                                                       const unsigned nblk, const unsigned noc, const int PH, const int PW,
                                                       const int PWP, const int PWH, const int ni,
                                                       const int8_t *__restrict__ zeros, const fastdiv_t dtx,
                                                       const fastdiv_t dty, const fastdiv_t dpwp, const unsigned out_bytes) {
    constexpr int TH = 16, BN = 128, WPX = 4, WOC = 4, STG = 3;
    extern __shared__ __attribute__((aligned(16))) int8_t dynlds[];
    uint8_t *slut = (uint8_t *)dynlds; // LDS byte address 0 (requant_pack LUT0)
    lds_base_must_be_zero(dynlds);
    const int nks = k64 / BK;
    int *dutab = (int *)(dynlds + LUTB);                     // [nks][4] unit offsets of the K chunks
    int *sbias = (int *)(dynlds + LUTB + ((nks * 16 + 255) & ~255)); // [BN]
    int8_t *wring = (int8_t *)(sbias + BN);                  // [STG][BN][64], rows swizzled like the ring tiles
    int8_t *patch = wring + STG * BN * BK;                   // ni x 8 KB

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv & 3, wn = wv >> 2;
    const unsigned id = xcd_remap(blockIdx.x, nblk);
    const unsigned t = id / noc;
    const int oc0 = (int)(id - t * noc) * BN;
    const int C = p.in_c, lgc = 31 - __builtin_clz((unsigned)C), cpp = C >> 4, lgcpp = lgc - 4;
    const int s = p.stride_w;
    const unsigned M = C >= 128 ? 6u : (C >= 64 ? 2u : 0u);

    if (HAS_LUT) {
        if (p.lut2) { if (tid < 128) ((uint32_t *)slut)[tid] = ((const uint32_t *)p.lut2)[tid]; }
        else if (tid < 64) ((uint32_t *)slut)[tid] = ((const uint32_t *)p.lut)[tid];
    }
    if (tid < BN) sbias[tid] = p.bias ? p.bias[oc0 + tid] : 0;
    const int rowbytes = p.kw * C, kbytes = p.kh * rowbytes;
    for (int i = tid; i < nks * 4; i += 512) {
        const int kb = i * 16;
        int du = 0; // K padding meets zero weights: any valid address
        if (kb < kbytes) {
            const int ky = kb / rowbytes, rem = kb - ky * rowbytes, kx = rem >> lgc, cc = (rem & (C - 1)) >> 4;
            const int dp = ky * PWP + (s == 2 ? (kx >> 1) + (kx & 1) * PWH : kx);
            du = dp * cpp + cc;
        }
        dutab[i] = du;
    }
    __syncthreads(); // tables visible (the K loop uses raw barriers)
    // tile, output offsets of this lane's pixels, residual operand (fetched first: the oldest vector-memory operations)
    const unsigned q0 = fdiv(t, dtx);
    const int tx = (int)(t - q0 * (unsigned)tiles_x);
    const unsigned f = fdiv(q0, dty);
    const int ty = (int)(q0 - f * (unsigned)tiles_y);
    const int frow = lane & 15, fchunk = lane >> 4;
    const int chan = wn * 64 + (lane >> 4) * (4 * WOC);
    const int pstride = p.out_pix_stride ? p.out_pix_stride : p.out_c;
    int voffs[WPX];
    uint32_t xw[WPX][WOC];
#pragma unroll
    for (int u = 0; u < WPX; u++) {
        const int oy = ty * TH + wm * WPX + u, ox = tx * PT_TW + frow;
        const unsigned off = f * (unsigned)p.out_stride + (unsigned)(oy * p.out_w + ox) * (unsigned)pstride +
                             (unsigned)(p.out_ch_off + oc0 + chan);
        const bool ok = oy < p.out_h && ox < p.out_w && oc0 + chan < p.out_c;
        voffs[u] = ok ? (int)off : -1;
#pragma unroll
        for (int q = 0; q < WOC; q++) xw[u][q] = 0;
        if (p.add && ok) {
            const v4i t4 = *(const v4i *)(p.add + off);
            xw[u][0] = t4[0]; xw[u][1] = t4[1]; xw[u][2] = t4[2]; xw[u][3] = t4[3];
        }
    }
    // input patch: instruction n of wave wv fills physical units (n*8 + wv)*64 + lane (unit = 16 bytes, swizzled)
    {
        const int iy0 = ty * TH * s - p.pad_top, ix0 = tx * PT_TW * s - p.pad_left;
        const int8_t *base = p.in + (size_t)f * p.in_stride + ((long)iy0 * p.in_w + ix0) * C;
#pragma unroll
        for (int n = 0; n < PWS_NIMAX; n++)
            if (n < ni) {
                const unsigned phys = (unsigned)((n * 8 + wv) * 64 + lane);
                const unsigned U = phys ^ ((phys >> 3) & M);
                const unsigned pp = U >> lgcpp, cc = U & (unsigned)(cpp - 1);
                const unsigned py = fdiv(pp, dpwp), col = pp - py * (unsigned)PWP;
                const int px = s == 2 ? ((int)col < PWH ? 2 * (int)col : 2 * ((int)col - PWH) + 1) : (int)col;
                const bool ok = (int)py < PH && px < PW && (unsigned)(iy0 + (int)py) < (unsigned)p.in_h &&
                                (unsigned)(ix0 + px) < (unsigned)p.in_w;
                glds16(ok ? base + ((long)py * p.in_w + px) * C + (int)cc * 16 : zeros, patch + (n * 8 + wv) * 1024);
            }
    }
    // weight ring: wave wv fetches rows oc0 + wv*16 .. +15 of a K step (one instruction per wave and step)
    const int schunk = (lane & 3) ^ (((lane >> 4) & 1) << 1);
    const int8_t *wsrc = p.w + (size_t)(oc0 + wv * 16 + (lane >> 2)) * k64 + schunk * 16;
    auto issue_w = [&](int ks) { glds16(wsrc + ks * BK, wring + (ks % STG) * (BN * BK) + wv * 16 * BK); };
    issue_w(0);
    if (nks > 1) issue_w(1);

    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)out_bytes, 0x00020000);
    const int lo = p.relu ? 0 : -128;
    const uint8_t *lut128 = slut + 128;
    int ubase[WPX]; // 16-byte unit of (tile row, column frow), tap (0,0), channel 0
#pragma unroll
    for (int u = 0; u < WPX; u++) ubase[u] = ((wm * WPX + u) * s * PWP + frow) * cpp;

    v4i acc[WOC][WPX];
    for (int ks = 0; ks < nks; ks++) {
        // K step ks has landed when at most the one younger step (ks + 1) is outstanding; the patch and the residual
        // operand are older than every weight step
        if (ks + 1 < nks) wait_vmcnt<1>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int du = dutab[ks * 4 + fchunk];
        const int8_t *ws = wring + (ks % STG) * (BN * BK);
        v4i xb[WPX], wa[WOC];
#pragma unroll
        for (int u = 0; u < WPX; u++) {
            const unsigned U = (unsigned)(ubase[u] + du);
            xb[u] = *(const v4i *)(patch + ((U ^ ((U >> 3) & M)) << 4));
        }
#pragma unroll
        for (int q = 0; q < WOC; q++) wa[q] = *(const v4i *)(ws + lds_off(wn * 64 + q * 16 + frow, fchunk));
        if (ks + 2 < nks) issue_w(ks + 2); // its slot was read in step ks - 1: every wave is past this step's barrier
        if (ks == 0) {
#pragma unroll
            for (int q = 0; q < WOC; q++) {
                const v4i b = *(const v4i *)(sbias + wn * 64 + q * 16 + (lane >> 4) * 4);
#pragma unroll
                for (int u = 0; u < WPX; u++) acc[q][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[q], xb[u], b, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < WOC; q++)
#pragma unroll
                for (int u = 0; u < WPX; u++) acc[q][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[q], xb[u], acc[q][u], 0, 0, 0);
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
        if (p.add) {
            const add_args_t ga = {p.add_s_conv, p.add_s_other, p.add_inv};
            if (fast) requant_pack<WOC * 4, HAS_LUT, true, true, true, true>(a, p.cs, lo, lut128, pk, xw[u], &ga);
            else requant_pack<WOC * 4, HAS_LUT, true, true, true>(a, p.cs, lo, lut128, pk, xw[u], &ga);
        } else {
            if (fast) requant_pack<WOC * 4, HAS_LUT, true, true, false, true>(a, p.cs, lo, lut128, pk);
            else requant_pack<WOC * 4, HAS_LUT, true, true>(a, p.cs, lo, lut128, pk);
        }
        __builtin_amdgcn_raw_buffer_store_b128((v4i){(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]}, orsrc, voffs[u], 0, 0);
    }
}

// ---------------------------------------------------------------------------------
// two-team strip kernel: the deep k x k stride-1 layers (3x3 with 64 / 128 / 256 input channels on the 80x80 .. 20x20
// maps).  What the counters say about the implicit-GEMM forms on these layers (profiles/r02_deep_sq_counters.md):
// the 8-wave tile is bound by L2 -> LDS DMA throughput (24 KB per K step at the ~21 B/clk/CU this access shape
// reaches), and conv_i8_patchw, which moves 2.3x fewer bytes, gains only 12 % because all workgroups of a round run in
// lockstep -- prologue, K loop and epilogue of the two workgroups on a CU coincide instead of overlapping.  This kernel:
//   * ONE 16-wave workgroup per CU = two TEAMS of 8 waves (4 pixel groups x 2 channel halves, 64 x 64 accumulators per
//     wave as before).  Each team owns a tile of 256 output pixels x 128 channels; tiles are runs of 256 FLAT pixels
//     (frame-major), so 40x40 and 20x20 maps fill every tile (a tile may straddle two frames: its input patch is then
//     two row segments).
//   * the teams run HALF A TILE apart: team 1 walks the K steps in rotated order (it starts at step H = nks / 2), so at
//     any moment both teams need the SAME 128 x 64 weight bytes -- one weight ring serves both (half the weight DMA of
//     two independent workgroups; int32 accumulation is order independent, so the rotation is exact), and one team's
//     epilogue / tile set-up falls into the middle of the other's K loop.
//   * the input patch of a tile is staged per 64-channel chunk (all taps of a chunk are served from LDS), double
//     buffered per team: the next chunk (or the next tile's first chunk) streams in, one DMA instruction per wave
//     and step, while the current one is computed.  Workgroups are persistent and walk a contiguous run of tiles.
//   * every wave issues the same deterministic sequence of vector-memory instructions per step, so one counted
//     s_waitcnt vmcnt(N) + one s_barrier per K step orders everything (loads, LDS-DMA and stores retire in order).
// LDS: [LUT 512][bias 512][weight ring 3 x 8 KB][team 0: 2 patch buffers][team 1: 2 patch buffers].
struct duo_args_t {
    int k64, taps, nchunk, nks, H, cH; // K steps (chunk-major: step = chunk * taps + tap); team 1 starts at step H, chunk cH
    int PWP, ni, patch_bytes;          // patch row pitch in positions; DMA instructions per wave and chunk; bytes per buffer
    unsigned total_pix, ntiles, noc, ngrp;
    fastdiv_t dhw, dow, dpwp;
    unsigned out_bytes, in_bytes, kw_magic;
};
__device__ __forceinline__ void wait_vmcnt_dyn(int n) { // n is wave-uniform; any immediate <= n is safe
    if (n == 1) wait_vmcnt<1>();       // the common case first: every compare + branch costs a scalar issue slot
    else if (n == 2) wait_vmcnt<2>();
    else if (n == 0) wait_vmcnt<0>();
    else if (n >= 6) wait_vmcnt<6>();
    else if (n == 5) wait_vmcnt<5>();
    else if (n == 4) wait_vmcnt<4>();
    else wait_vmcnt<3>();
}
// Scalar instructions are the scarce resource of a 16-wave workgroup (one scalar ALU per CU: 16 waves x ~120 scalar
// instructions per K step made the first version of this kernel scalar-bound at 3500 cycles per step).  So the K loop
// runs in EPOCHS of TAPS = 9 steps (one 64-channel chunk of a 3x3 kernel) with the step body unrolled: tap index,
// ring slot and the tap's offset inside the patch are compile-time constants, all tile / epoch bookkeeping happens
// once per epoch, and a step carries ~20 scalar instructions.
template <bool HAS_LUT, bool HAS_ADD>
__global__ __launch_bounds__(1024) void conv_i8_duo(const mhip_conv_i8_t p, const duo_args_t g) {
    constexpr int BN = 128, WPX = 4, WOC = 4, STG = 3, P = 256, TAPS = 9;
    extern __shared__ __attribute__((aligned(16))) int8_t dynlds[];
    uint8_t *slut = (uint8_t *)dynlds; // LDS byte address 0 (requant_pack LUT0)
    lds_base_must_be_zero(dynlds);
    int *sbias = (int *)(dynlds + LUTB);
    int *pslots = (int *)(dynlds + LUTB + BN * 4); // 16 waves x 8 ints: descriptor of the patch epoch a wave prefetches
    int8_t *wring = dynlds + LUTB + BN * 4 + 512;
    int8_t *patches = wring + STG * BN * BK;
    // Scalar registers are what this kernel runs out of (106 of 106 in use made the compiler park uniform values in
    // vector registers and spill those: every reload then dragged an s_waitcnt vmcnt(0) into the K loop).  Everything
    // that is NOT needed in every step -- tile set-up, patch prefetch, epilogue -- therefore re-reads its parameters from
    // the kernel-argument segment through a pointer the optimiser cannot see through, instead of keeping them live.
    // ... and per-lane values derived from the lane id are recomputed there from an opaque copy of it: loop-invariant
    // code motion otherwise hoists that arithmetic out of the epoch loop and keeps (spills) its results across it
    auto lane_ = [&]() __attribute__((always_inline)) {
        int l = (int)(threadIdx.x & 63u);
        asm volatile("" : "+v"(l));
        return l;
    };
    typedef const __attribute__((address_space(4))) char *kaptr_t;
    auto P_ = [&]() __attribute__((always_inline)) {
        kaptr_t k = (kaptr_t)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(k));
        return (const __attribute__((address_space(4))) mhip_conv_i8_t *)k;
    };
    auto G_ = [&]() __attribute__((always_inline)) {
        kaptr_t k = (kaptr_t)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(k));
        return (const __attribute__((address_space(4))) duo_args_t *)(k + ((sizeof(mhip_conv_i8_t) + 7) & ~(size_t)7));
    };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // waves go to SIMDs round-robin, so waves w, w+4, w+8, w+12 share one: two of each team
    const int team = (wv >> 2) & 1, tw = (wv & 3) | ((wv >> 3) << 2);
    const int wm = tw & 3, wn = tw >> 2;
    int8_t *const mypatch = patches + team * 2 * g.patch_bytes;

    const unsigned id = xcd_remap(blockIdx.x, g.noc * g.ngrp);
    const unsigned grp = id / g.noc;
    const int oc0 = (int)(id - grp * g.noc) * BN;
    const unsigned t0 = (unsigned)(((unsigned long long)grp * g.ntiles) / g.ngrp);
    const unsigned t1 = (unsigned)(((unsigned long long)(grp + 1) * g.ntiles) / g.ngrp);
    if (t0 >= t1) return; // uniform for the workgroup
    const int nchunk = g.nchunk, nks = g.nks;
    const int n0 = (int)((t1 - t0 + 1) >> 1), n1 = (int)((t1 - t0) >> 1); // tiles of team 0 (t0, t0+2, ..) and team 1 (t0+1, ..)
    const int G0 = n0 * nks, G1 = n1 ? g.H + n1 * nks : 0;
    const int G = G0 > G1 ? G0 : G1;                 // steps until both teams have finished
    const int NE = (G + TAPS - 1) / TAPS;            // epochs (the last one may be partial: its idle steps only pass barriers)
    const int nmine = team ? n1 : n0;
    const int gstart = team ? g.H : 0;               // first step at which this team works
    const int startc = team ? g.cH : 0;              // ... the chunk its tiles start with
    const int kstart = gstart % TAPS;                // ... and the tap (0, or 4 for team 1 of a one-chunk layer)
    const int kend = kstart == 0 ? TAPS - 1 : kstart - 1;

    if (HAS_LUT) {
        if (p.lut2) { if (tid < 128) ((uint32_t *)slut)[tid] = ((const uint32_t *)p.lut2)[tid]; }
        else if (tid < 64) ((uint32_t *)slut)[tid] = ((const uint32_t *)p.lut)[tid];
    }
    if (tid < BN) sbias[tid] = p.bias ? p.bias[oc0 + tid] : 0;

    if (G_()->nks != g.nks || P_()->in_c != p.in_c) __builtin_trap(); // the kernel-argument offsets assumed above
    const int C = p.in_c;
    const int frow = lane & 15, fchunk = lane >> 4;
    const uint8_t *lut128 = slut + 128;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, p.oc_pad * g.k64, 0x00020000);

    // the row segments of a tile (uniform): flat pixels [tile*P, tile*P + P) touch one frame (segment A only) or two.
    // 3x3, stride 1, pad 1 (host-checked): segment rows = output rows + 2, first input row = first output row - 1.
    struct seg_t { unsigned fA, fB; int oyA0, nA, nB; };
    auto segments = [&](unsigned tile) __attribute__((always_inline)) {
        const auto *G = G_();
        const auto *Pp = P_();
        seg_t s;
        const unsigned hw = (unsigned)(Pp->out_h * Pp->out_w), tp = G->total_pix;
        const unsigned g0 = tile * P, gl = (g0 + P < tp ? g0 + P : tp) - 1u;
        const fastdiv_t dhw = {G->dhw.m, G->dhw.s1, G->dhw.s2}, dow = {G->dow.m, G->dow.s1, G->dow.s2};
        s.fA = fdiv(g0, dhw);
        s.fB = fdiv(gl, dhw);
        s.oyA0 = (int)fdiv(g0 - s.fA * hw, dow);
        const int oyB1 = (int)fdiv(gl - s.fB * hw, dow);
        const bool two = s.fA != s.fB;
        const int oyA1 = two ? Pp->out_h - 1 : oyB1;
        s.nA = oyA1 - s.oyA0 + 3;
        s.nB = two ? oyB1 + 3 : 0;
        return s;
    };
    // descriptor of a patch epoch in this wave's LDS slot: {fA * in_stride, fB * in_stride, first input row of A, rows of A,
    // rows of B, chunk * 64}.  Written once per epoch (scalar arithmetic), read back into VECTOR registers by every piece.
    int *const myslot = pslots + wv * 8;
    auto put_epoch = [&](unsigned tile, int chunk64) __attribute__((always_inline)) {
        const seg_t s = segments(tile);
        const unsigned istr = (unsigned)P_()->in_stride;
        if (lane == 0) {
            *(v4i *)myslot = (v4i){(int)(s.fA * istr), (int)(s.fB * istr), s.oyA0 - 1, s.nA};
            *(v4i *)(myslot + 4) = (v4i){s.nB, chunk64, 0, 0};
        }
    };
    // One DMA instruction of a patch chunk: instruction n of team wave tw fills 16-byte units (n*8 + tw)*64 + lane of the
    // buffer (unit = position * 4 + 16-byte piece of the 64-channel chunk, swizzled as in conv_i8_patch for C = 64).
    // This lane's unit of instruction n is position pos0 + n*128, piece ccoff / 16: both fixed for the whole run.
    auto issue_patch = [&](int n, int8_t *dst) __attribute__((always_inline)) {
        const auto *G = G_();
        const auto *Pp = P_();
        const v4i d0 = *(const v4i *)myslot, d1 = *(const v4i *)(myslot + 4);
        const unsigned ln = (unsigned)lane_();
        const unsigned lane_sw = ln ^ ((ln >> 3) & 2u);
        const unsigned pos = (unsigned)tw * 16u + (lane_sw >> 2) + (unsigned)n * 128u, ccoff = (lane_sw & 3u) * 16u;
        const fastdiv_t dpwp = {G->dpwp.m, G->dpwp.s1, G->dpwp.s2};
        const unsigned prow = fdiv(pos, dpwp), px = pos - prow * (unsigned)G->PWP;
        const int nA = d0[3], nB = d1[0];
        const bool inA = (int)prow < nA;
        const int iy = inA ? d0[2] + (int)prow : (int)prow - nA - 1;
        const int ix = (int)px - 1;
        const int in_w = Pp->in_w;
        const bool ok = (inA || (int)prow - nA < nB) && (unsigned)iy < (unsigned)Pp->in_h && (unsigned)ix < (unsigned)in_w;
        const unsigned off = (unsigned)(inA ? d0[0] : d0[1]) + (unsigned)((iy * in_w + ix) * Pp->in_c + d1[1]) + ccoff;
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)Pp->in, 0, (int)G->in_bytes, 0x00020000);
        blds16(xrs, ok ? (int)off : -1, 0, dst + (n * 8 + tw) * 1024);
    };
    // weights of one K step (chunk, tap): bytes [tap * C + chunk * 64, + 64) of every row; each of the 16 waves fetches
    // 8 rows (lanes 0..31), the same source-side swizzle as the ring kernels
    const int schunk = (lane & 3) ^ (((lane >> 4) & 1) << 1);
    const int wvoff = (oc0 + wv * 8 + ((lane >> 2) & 7)) * g.k64 + schunk * 16;

    // ---- per-team tile state
    int ubase[WPX];
    v4i acc[WOC][WPX];
    auto out_off = [&](unsigned tile, int u) __attribute__((always_inline)) { // output byte offset of this lane's pixel of subtile u, or -1
        const auto *G = G_();
        const auto *Pp = P_();
        const int ln = lane_();
        const unsigned gp = tile * P + (unsigned)((wm * WPX + u) * 16 + (ln & 15));
        const bool valid = gp < G->total_pix;
        const unsigned gq = valid ? gp : 0u;
        const unsigned hw = (unsigned)(Pp->out_h * Pp->out_w);
        const fastdiv_t dhw = {G->dhw.m, G->dhw.s1, G->dhw.s2};
        const unsigned f = fdiv(gq, dhw), rem = gq - f * hw;
        const int out_c = Pp->out_c, pstride = Pp->out_pix_stride ? Pp->out_pix_stride : out_c;
        const int ch = oc0 + wn * 64 + (ln >> 4) * (4 * WOC);
        const unsigned off = f * (unsigned)Pp->out_stride + rem * (unsigned)pstride + (unsigned)(Pp->out_ch_off + ch);
        return valid && ch < out_c ? (int)off : -1;
    };
    auto setup_tile = [&](unsigned tile) __attribute__((always_inline)) {
        const seg_t s = segments(tile);
        const auto *G = G_();
        const auto *Pp = P_();
        const unsigned hw = (unsigned)(Pp->out_h * Pp->out_w);
        const int out_w = Pp->out_w, pwp = G->PWP;
        const fastdiv_t dhw = {G->dhw.m, G->dhw.s1, G->dhw.s2}, dow = {G->dow.m, G->dow.s1, G->dow.s2};
        const int ln = lane_(), fchunk = ln >> 4;
#pragma unroll
        for (int u = 0; u < WPX; u++) {
            const unsigned gp = tile * P + (unsigned)((wm * WPX + u) * 16 + (ln & 15));
            const bool valid = gp < G->total_pix;
            const unsigned gq = valid ? gp : 0u;
            const unsigned f = fdiv(gq, dhw), rem = gq - f * hw;
            const int oy = (int)fdiv(rem, dow), ox = (int)rem - oy * out_w;
            const int prow = f == s.fA ? oy - s.oyA0 : s.nA + oy;
            ubase[u] = valid ? (prow * pwp + ox) * 4 + fchunk : fchunk;
        }
        // accumulators start at the bias (lane holds channels wn*64 + q*16 + (lane>>4)*4 .. +3 of every pixel subtile)
#pragma unroll
        for (int q = 0; q < WOC; q++) {
            const v4i b = *(const v4i *)(sbias + wn * 64 + q * 16 + (ln >> 4) * 4);
#pragma unroll
            for (int u = 0; u < WPX; u++) acc[q][u] = b;
        }
    };

    // ---- prologue: the first patch epoch of each team, then the weights of steps 0 and 1
    __syncthreads(); // LUT / bias visible
    unsigned mytile = t0 + (unsigned)team;
    if (nmine > 0) {
        put_epoch(mytile, startc * 64);
        for (int n = 0; n < g.ni; n++) issue_patch(n, mypatch);
    }
    if (lane < 32) blds16(wrs, wvoff, 0, wring + wv * 512);                              // step 0: chunk 0, tap 0
    if (G > 1 && lane < 32) blds16(wrs, wvoff + C, 0, wring + BN * BK + wv * 512);       // step 1: chunk 0, tap 1

    bool working = false, pvalid = false, have = false;
    int lc = 0, ebuf = 0, tdone = 0; // lc: chunks of the current tile done so far (in this team's order)
    int8_t *pdst = mypatch;
    const int8_t *pb = mypatch;
    const int rowoff1 = g.PWP * 4, rowoff2 = g.PWP * 8; // 16-byte units per patch row (ky = 1, 2)
    int gs = 0, kc = 0; // global step and the chunk of the current epoch
    int kc64 = 0, kn64 = nchunk > 1 ? 64 : 0; // chunk * 64 of this epoch and of the next
    // vmcnt bookkeeping (per wave, in issue order): the wait of step gs must see the weight piece this wave issued in
    // step gs-1 landed (the weights of step gs+1).  Younger than that piece: what the wave issued between it and this
    // step's weight piece (va), this step's weight piece, and what it has issued since (vx)
    int va = 0, vx = 0;
    v4i xb[WPX], wa[WOC];

    // The two teams work in opposite PHASES: while one reads its MFMA operands from LDS and does its bookkeeping (LOAD),
    // the other runs its 16 MFMAs per wave (COMPUTE).  Both execute the same code, LOAD(gs) then COMPUTE(gs), but the
    // step's barrier sits at a different place: team 0 meets it AFTER its compute, team 1 BETWEEN its load and its
    // compute.  Between two rendezvous team 0 therefore does [weights DMA, LOAD(gs+1), COMPUTE(gs+1)] and team 1
    // [COMPUTE(gs), weights DMA, LOAD(gs+1)]: one team's MFMAs always run beside the other's load.  (With every wave in
    // the same phase -- the first version -- the matrix pipe idled through everyone's load: 38 % busy.)
    auto load = [&](auto KTc) __attribute__((always_inline)) {
        constexpr int KT = decltype(KTc)::value;
        if ((KT == 0 || KT == 4) && KT == kstart) { // a patch epoch of this team begins
            if (!working) {
                if (gs == gstart && nmine > 0) { working = true; lc = 0; }
            } else {
                ebuf ^= 1; // the prefetch filled the other buffer
            }
            if (working) {
                pb = mypatch + ebuf * g.patch_bytes;
                pdst = mypatch + (ebuf ^ 1) * g.patch_bytes;
                if (lc == 0) setup_tile(mytile);
                // the epoch to prefetch: the next chunk of this tile, or the first chunk of this team's next tile
                const unsigned ptile = lc + 1 == nchunk ? mytile + 2 : mytile;
                pvalid = ptile < t1;
                if (pvalid) put_epoch(ptile, kn64);
            }
        }
        if (working) { // fragment reads of this wave's 16 MFMAs
            constexpr int KY = KT / 3, KX = KT % 3;
            const int du = (KY == 0 ? 0 : (KY == 1 ? rowoff1 : rowoff2)) + KX * 4;
            const int8_t *ws = wring + (KT % STG) * (BN * BK) + lds_off(wn * 64 + frow, fchunk);
#pragma unroll
            for (int u = 0; u < WPX; u++) {
                const unsigned U = (unsigned)(ubase[u] + du);
                xb[u] = *(const v4i *)(pb + ((U ^ ((U >> 3) & 2u)) << 4));
            }
#pragma unroll
            for (int q = 0; q < WOC; q++) wa[q] = *(const v4i *)(ws + q * 16 * BK); // 16 rows further the swizzle repeats
            have = true;
            // the data must be in registers before the barrier: after it the other waves may start DMA into these bytes
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    auto compute = [&](auto KTc) __attribute__((always_inline)) { // KT = the tap of the step whose operands were loaded
        constexpr int KT = decltype(KTc)::value;
        if (!have) return;
        have = false;
#pragma unroll
        for (int q = 0; q < WOC; q++)
#pragma unroll
            for (int u = 0; u < WPX; u++) acc[q][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[q], xb[u], acc[q][u], 0, 0, 0);
        // one piece of the next epoch's patch per step (first ni steps of an epoch), after the MFMAs have been issued: its
        // address arithmetic then reuses the registers of the fragments instead of competing with them
        __builtin_amdgcn_sched_barrier(0);
        if (pvalid) {
            int n = KT - kstart;
            if (n < 0) n += TAPS;
            asm volatile("" : "+s"(n)); // opaque: otherwise the per-piece lane arithmetic of all nine steps is hoisted out of
                                        // the epoch loop and its results are kept (and spilled) across it
            if (n < g.ni) {
                issue_patch(n, pdst);
                vx += 1;
            }
        }
        if ((KT == TAPS - 1 || KT == 3) && KT == kend) { // a patch epoch of this team ends
            if (++lc == nchunk) { // ... and with it the tile: requantise, LUT, store
                const auto *Pp = P_();
                const float cs = Pp->cs;
                const int lo = Pp->relu ? 0 : -128;
                const bool fast = HAS_LUT && Pp->lut2 != nullptr;
                const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)Pp->out, 0, (int)G_()->out_bytes, 0x00020000);
#pragma unroll
                for (int u = 0; u < WPX; u++) {
                    uint32_t pk[WOC];
                    int a[WOC * 4];
#pragma unroll
                    for (int q = 0; q < WOC; q++)
#pragma unroll
                        for (int r = 0; r < 4; r++) a[q * 4 + r] = acc[q][u][r];
                    if (fast) requant_pack<WOC * 4, HAS_LUT, true, true, false, true>(a, cs, lo, lut128, pk);
                    else requant_pack<WOC * 4, HAS_LUT, true, true>(a, cs, lo, lut128, pk);
                    __builtin_amdgcn_raw_buffer_store_b128((v4i){(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]}, orsrc, out_off(mytile, u), 0, 0);
                }
                vx += WPX;
                lc = 0;
                mytile += 2;
                if (++tdone >= nmine) { working = false; pvalid = false; } // (ebuf flips where the next tile begins)
            }
        }
    };
    auto step = [&](auto KTc) __attribute__((always_inline)) {
        constexpr int KT = decltype(KTc)::value;
        // every wave fetches its piece of the weights of step gs + 2 (into the ring slot step gs - 1 used)
        const bool more = gs + 2 < G;
        if (more) {
            constexpr int T2 = (KT + 2) % TAPS, SLOT = (KT + 2) % STG; // epochs are 9 steps: slot = tap mod 3
            const int ck = KT + 2 >= TAPS ? kn64 : kc64;
            if (lane < 32) blds16(wrs, wvoff + T2 * C + ck, 0, wring + SLOT * (BN * BK) + wv * 512);
            va = vx;
            vx = 0;
        }
        load(KTc);
        if (team == 1) {
            wait_vmcnt_dyn(more ? va + 1 + vx : 0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        compute(KTc);
        if (team == 0) {
            wait_vmcnt_dyn(more ? va + 1 + vx : 0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        gs++;
    };

    wait_vmcnt<1>(); // the patch and the weights of step 0 (only the piece of step 1 may still be in flight) ...
    __builtin_amdgcn_s_barrier(); // ... of every wave
    asm volatile("" ::: "memory");
    for (int e = 0; e < NE; e++) {
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{});
        step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{});
        step(std::integral_constant<int, 7>{});
        step(std::integral_constant<int, 8>{});
        kc = kc + 1 == nchunk ? 0 : kc + 1;
        kc64 = kc * 64;
        kn64 = (kc + 1 == nchunk ? 0 : kc + 1) * 64;
    }
    wait_vmcnt<0>(); // nothing of this workgroup may still be writing LDS when it ends
}

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


// ---------------------------------------------------------------------------------
// small-channel kernel (in_c <= 4, kw <= 8: the RGB stem).  The input patch of an
// 8x16 output tile is staged ONCE in LDS with every pixel widened to 4 bytes, so a
// kernel row of a pixel is 32 contiguous LDS bytes (kw*4 used, the rest meets zero
// weights) and one MFMA K step covers two kernel rows.  Weights ([oc][kh][32]) stay in
// LDS for the lifetime of the (persistent) workgroup.  Input bytes are read once.
#define SC_TH 16
#define SC_TW 16
#define SC_BP (SC_TH * SC_TW)
template <int WOC, bool HOT = false>
__global__ __launch_bounds__(NTHREADS) void conv_i8_smallc(const mhip_conv_i8_t p, const int k64, const int tiles_x,
                                                           const int tiles_y, const unsigned ntiles_all, const int PH,
                                                           const int PW, const int PWp, const fastdiv_t dhw,
                                                           const fastdiv_t dtx, const fastdiv_t dty, const fastdiv_t dgpr,
                                                           const int tile_bytes) {
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
    for (int i = tid; i < BN * (k64 / 16); i += NTHREADS) {
        const int row = i / (k64 / 16), c = i - row * (k64 / 16);
        *(v4i *)(wl + (c >> 2) * (BN * BK) + lds_off(row, c & 3)) = *(const v4i *)(p.w + (size_t)row * k64 + c * 16);
    }
    for (int i = tid; i < 2 * patch_bytes / 4; i += NTHREADS) ((uint32_t *)patch0)[i] = 0;
    if (tid < BN) ((int *)sbias)[tid] = p.bias ? p.bias[tid] : 0;

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
    const bool fast3 = HOT || (p.in_c == 3 && p.in_w >= 4);
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
                __builtin_memcpy(&pre[j], src + ((long)iyc * p.in_w + ixc) * 3, 16); // unaligned dwordx4, 12 bytes used
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
                v4i l;
                l[0] = (int)(d0 & 0xFFFFFFu);
                l[1] = (int)(((d0 >> 24) | (d1 << 8)) & 0xFFFFFFu);
                l[2] = (int)(((d1 >> 16) | (d2 << 16)) & 0xFFFFFFu);
                l[3] = (int)(d2 >> 8);
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
            const bool rok = chok && oy < p.out_h; // stores always issue: the same vmcnt in every wave
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
}

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

// launch policy knobs.  Defaults are the measured optimum on MI355X; the environment (read once) and
// mhip_conv_i8_tune() (tests: force the multi-tile walk on small inputs) override them.
struct tune_t {
    int init;
    int persist;        // MARS_HIP_PERSIST       1: persistent kernel where eligible
    int persist_stages; // MARS_HIP_PSTAGES       ring depth of the persistent kernel (2 | 3)
    int persist_maxk;   // MARS_HIP_PERSIST_MAXK  deepest K loop (64-byte steps) that still walks tiles
    int persist_slots;  // MARS_HIP_PSLOTS        0: what the device holds at once, else this many workgroups
    int stages;         // MARS_HIP_STAGES        0: auto, else ring depth of the one-tile kernel (2 | 3 | 4)
    int bpx;            // MARS_HIP_BPX           0: auto, else pixels per workgroup (128 | 256)
    int variant;        // MARS_HIP_VARIANT       0: policy, else this launch variant wherever the layer allows it (tests)
    int bufmode;        // MARS_HIP_BUFMODE       1: buffer-addressed K loop where eligible
    int small_batch;    // MARS_HIP_SMALL_BATCH   1: launches with few workgroups take the small-tile policy (default_variant)
    int rgb_direct;     // MARS_HIP_RGB_DIRECT    1: the RGB stem runs in its operand-direct form (conv_i8_rgb) where eligible
    int wres;           // MARS_HIP_WRES          bit 0 / 1: the default policy may keep the weights resident in LDS (tile
                        //                        walker) for single / paired launches
};
static tune_t g_tune;
static int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
static const tune_t &tune() {
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
        g_tune.init = 1;
    }
    return g_tune;
}
extern "C" int mhip_conv_i8_tune(const char *key, int value) {
    (void)tune();
    struct { const char *k; int *v; } tab[] = {{"persist", &g_tune.persist}, {"persist_stages", &g_tune.persist_stages},
                                               {"persist_maxk", &g_tune.persist_maxk}, {"persist_slots", &g_tune.persist_slots}, {"wres", &g_tune.wres}, {"rgb_direct", &g_tune.rgb_direct}, {"small_batch", &g_tune.small_batch},
                                               {"stages", &g_tune.stages}, {"bpx", &g_tune.bpx}, {"variant", &g_tune.variant}, {"bufmode", &g_tune.bufmode}};
    for (auto &e : tab)
        if (key && !strcmp(key, e.k)) {
            *e.v = value;
            return 0;
        }
    return -1;
}

static long persist_out_bytes(const mhip_conv_i8_t *p);
static long in_extent_bytes(const mhip_conv_i8_t *p);
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

// bytes from p->in to the end of the last frame's pixels (the buffer resource's range)
static long in_extent_bytes(const mhip_conv_i8_t *p) {
    return (long)(p->frames - 1) * (long)p->in_stride + (long)p->in_h * p->in_w * p->in_c;
}
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

// bytes from p->out to the end of the last pixel row the layer can write
static long persist_out_bytes(const mhip_conv_i8_t *p) {
    const long pstride = p->out_pix_stride ? p->out_pix_stride : p->out_c;
    return (long)(p->frames - 1) * (long)p->out_stride + (long)p->out_h * p->out_w * pstride;
}

// LDS of the weights-resident form: LUT + 2 pixel-tile stages + every K step of one channel tile
template <int BPX, int BN>
static size_t wres_lds(int k64) { return LUTB + 2 * (size_t)BPX * BK + (size_t)(k64 / BK) * BN * BK; }

template <int BPX, int BN, int STAGES, bool HAS_LUT, bool SEG = false, bool PAIR = false>
static int launch_persist_t(const mhip_conv_i8_t *p, long total_pix, int k64, int lg, unsigned magic,
                            const mhip_conv_i8_t *second = nullptr, int wres = 0) {
    const unsigned noc0 = (unsigned)(p->oc_pad / BN);
    const unsigned npt = (unsigned)((total_pix + BPX - 1) / BPX), noc = noc0 + (PAIR ? (unsigned)(second->oc_pad / BN) : 0u);
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
//   code = 16: input patch staged once, weights streamed (8 waves, 16 x 16 pixels x 128 channels)
//   code = 17: two-team strip kernel (16 waves: two 256-pixel x 128-channel tiles half a tile apart, shared weight ring)
//   code = 18 / 19: 128-byte K steps (whole-line DMA requests), 128 x 128 tile on 4 waves / 256 x 128 tile on 8 waves
#define NVARIANTS 20
struct variant_t {
    int persist, bpx, stages, patch, ks2, w8, wres, pws, duo, r128;
};
static int variant_code(const variant_t &v) {
    if (v.r128) return 17 + v.r128;
    if (v.duo) return 17;
    if (v.pws) return 16;
    if (v.wres) return v.bpx == 256 ? 15 : 14;
    if (v.w8) return 13;
    if (v.ks2) return 12;
    if (v.patch) return v.patch == 16 ? 10 : (v.patch == 8 ? 9 : 11);
    return 1 + (v.persist ? 1 : 0) + (v.bpx == 256 ? 2 : 0) + (v.stages == 3 ? 4 : 0);
}
static variant_t variant_of(int code) {
    if (code >= 18 && code <= 20) return variant_t{0, 0, 0, 0, 0, 0, 0, 0, 0, code - 17};
    if (code == 17) return variant_t{0, 0, 0, 0, 0, 0, 0, 0, 1, 0};
    if (code == 16) return variant_t{0, 0, 0, 0, 0, 0, 0, 1, 0, 0};
    if (code == 14 || code == 15) return variant_t{1, code == 15 ? 256 : 128, 2, 0, 0, 0, 1, 0, 0, 0};
    if (code == 13) return variant_t{0, 256, 3, 0, 0, 1, 0, 0, 0, 0};
    if (code == 12) return variant_t{0, 128, 2, 0, 1, 0, 0, 0, 0, 0};
    if (code >= 9) return variant_t{0, 0, 0, code == 10 ? 16 : (code == 9 ? 8 : 4), 0, 0, 0, 0, 0, 0};
    const int c = code - 1;
    return variant_t{c & 1, (c & 2) ? 256 : 128, (c & 4) ? 3 : 2, 0, 0, 0, 0, 0, 0, 0};
}

// ---- patch-staged kernel: geometry, eligibility, launch
struct patch_geom_t {
    int bn, tiles_x, tiles_y, PH, PW, PWP, PWH, ni, nks, dbl;
    size_t lds;
};
static bool patch_geom(const mhip_conv_i8_t *p, int th, patch_geom_t *g) {
    const bool direct = !p->out_nchw && ((p->out_c | p->out_pix_stride | p->out_ch_off) & 15) == 0;
    const int C = p->in_c, s = p->stride_w;
    if (!direct || !p->safe || (C != 32 && C != 64 && C != 128) || (s != 1 && s != 2) || p->stride_h != s ||
        p->kh > 7 || p->kw > 7 || p->kh * p->kw < 2 || p->row_pad != p->kw * C || persist_out_bytes(p) > 0x7fffffffL)
        return false;
    const int k64 = (p->kh * p->row_pad + BK - 1) / BK * BK;
    g->nks = k64 / BK;
    g->bn = p->oc_pad % 64 == 0 ? 64 : 32;
    g->tiles_x = (p->out_w + PT_TW - 1) / PT_TW;
    g->tiles_y = (p->out_h + th - 1) / th;
    // mostly full tiles only: a tile computes th x 16 pixels whether the image has them or not
    if ((double)p->out_h * p->out_w < 0.85 * (double)g->tiles_x * PT_TW * g->tiles_y * th) return false;
    g->PH = (th - 1) * s + p->kh;
    g->PW = (PT_TW - 1) * s + p->kw;
    g->PWH = s == 2 ? (g->PW + 1) / 2 : 0;
    g->PWP = s == 2 ? 2 * g->PWH : g->PW;
    const long units = (long)g->PH * g->PWP * (C / 16);
    g->ni = (int)((units + 255) / 256);
    if (g->ni > PT_NIMAX) return false;
    // two workgroups per CU in any case (80 KB each): double-buffered patch if that fits, else one buffer
    const bool pre = p->pre_w != nullptr; // + the 1x1's table, its weights and the patch of its output
    if (pre && (s != 1 || (C != 32 && C != 64) || !p->pre_bias || !p->pre_lut2 || !p->lut2)) return false;
    const size_t fixed = LUTB + (pre ? 512 + (size_t)C * BK + (size_t)g->ni * 4096 : 0) +
                         (((size_t)g->nks * 16 + 255) & ~(size_t)255) + (size_t)g->nks * g->bn * BK;
    g->dbl = fixed + 2 * (size_t)g->ni * 4096 <= 80 * 1024;
    g->lds = fixed + (g->dbl ? 2 : 1) * (size_t)g->ni * 4096;
    if (g->lds > 80 * 1024) return false;
    if ((long)g->tiles_x * g->tiles_y * p->frames > 0x7fffffffL) return false;
    return true;
}

template <int TH, int BN, bool HAS_LUT, bool PRE = false>
static int launch_patch_t(const mhip_conv_i8_t *p, int k64, const patch_geom_t &g) {
    auto kern = conv_i8_patch<TH, BN, HAS_LUT, PRE>;
    // workgroups the device holds at once at THIS layer's LDS size (small patches fit 3-4 per CU), cached per size
    static int cus = 0;
    static size_t slots_lds[4];
    static int slots_n[4], nslots = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess ||
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
        if (nslots < 4) { slots_lds[nslots] = g.lds; slots_n[nslots++] = slots; }
    }
    const unsigned ntiles = (unsigned)((long)g.tiles_x * g.tiles_y * p->frames);
    const unsigned noc = (unsigned)(p->oc_pad / BN);
    unsigned gx = (unsigned)(tune().persist_slots > 0 ? tune().persist_slots : slots) / noc;
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    const int xmap = gx >= 8 && ntiles < 0x0fffffffu; // ids reach 8 x the longest range
    if (xmap) gx &= ~7u;
    hipLaunchKernelGGL(kern, dim3(gx, noc), dim3(NTHREADS), g.lds, mhip_stream_native(), *p, k64,
                       g.tiles_x, g.tiles_y, ntiles, g.PH, g.PW, g.PWP, g.PWH, g.ni, (const int8_t *)mhip_zero_page(),
                       make_fastdiv((unsigned)g.tiles_x), make_fastdiv((unsigned)g.tiles_y), make_fastdiv((unsigned)g.PWP),
                       (unsigned)persist_out_bytes(p), g.dbl, xmap);
    return mhip_check(hipGetLastError(), "conv_i8_patch launch");
}

// ---- patch-staged input with streamed weights (conv_i8_patchw): geometry, eligibility, launch
struct pws_geom_t {
    int tiles_x, tiles_y, PH, PW, PWP, PWH, ni;
    size_t lds;
};
static bool pws_geom(const mhip_conv_i8_t *p, pws_geom_t *g) {
    const bool direct = !p->out_nchw && ((p->out_c | p->out_pix_stride | p->out_ch_off) & 15) == 0;
    const int C = p->in_c, s = p->stride_w;
    if (!direct || !p->safe || (C != 64 && C != 128) || (s != 1 && s != 2) || p->stride_h != s || p->kh > 7 || p->kw > 7 ||
        p->kh * p->kw < 2 || p->row_pad != p->kw * C || p->oc_pad % 128 != 0 || persist_out_bytes(p) > 0x7fffffffL || p->nseg > 1)
        return false;
    const int k64 = (p->kh * p->row_pad + BK - 1) / BK * BK, nks = k64 / BK;
    if (nks < 3) return false;
    g->tiles_x = (p->out_w + PT_TW - 1) / PT_TW;
    g->tiles_y = (p->out_h + 15) / 16;
    // mostly full tiles only: a tile computes 16 x 16 pixels whether the map has them or not (40 x 40 maps fill 69 %:
    // measured 90 vs 80 us against the 8-wave implicit-GEMM tile; 48 x 48 maps fill 100 %: 92 vs 105 us)
    if ((double)p->out_h * p->out_w < 0.85 * (double)g->tiles_x * PT_TW * g->tiles_y * 16) return false;
    g->PH = 15 * s + p->kh;
    g->PW = (PT_TW - 1) * s + p->kw;
    g->PWH = s == 2 ? (g->PW + 1) / 2 : 0;
    g->PWP = s == 2 ? 2 * g->PWH : g->PW;
    const long units = (long)g->PH * g->PWP * (C / 16);
    g->ni = (int)((units + 511) / 512);
    if (g->ni > PWS_NIMAX) return false;
    g->lds = LUTB + (((size_t)nks * 16 + 255) & ~(size_t)255) + 128 * 4 + 3 * (size_t)128 * BK + (size_t)g->ni * 8192;
    if (g->lds > 80 * 1024) return false;
    if ((long)g->tiles_x * g->tiles_y * p->frames * (p->oc_pad / 128) > 0x7fffffffL) return false;
    return true;
}
template <bool HAS_LUT>
static int launch_pws_t(const mhip_conv_i8_t *p, int k64, const pws_geom_t &g) {
    static bool attr = false;
    if (!attr && hipFuncSetAttribute((const void *)conv_i8_patchw<HAS_LUT>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)
        return mhip_check(hipErrorUnknown, "conv_i8_patchw LDS attribute");
    attr = true;
    const unsigned ntiles = (unsigned)((long)g.tiles_x * g.tiles_y * p->frames), noc = (unsigned)(p->oc_pad / 128);
    hipLaunchKernelGGL((conv_i8_patchw<HAS_LUT>), dim3(ntiles * noc), dim3(512), g.lds, mhip_stream_native(), *p, k64, g.tiles_x,
                       g.tiles_y, ntiles * noc, noc, g.PH, g.PW, g.PWP, g.PWH, g.ni, (const int8_t *)mhip_zero_page(),
                       make_fastdiv((unsigned)g.tiles_x), make_fastdiv((unsigned)g.tiles_y), make_fastdiv((unsigned)g.PWP),
                       (unsigned)persist_out_bytes(p));
    return mhip_check(hipGetLastError(), "conv_i8_patchw launch");
}
static int launch_pws(const mhip_conv_i8_t *p, int k64) {
    pws_geom_t g;
    if (!pws_geom(p, &g)) return -1;
    return p->lut ? launch_pws_t<true>(p, k64, g) : launch_pws_t<false>(p, k64, g);
}

// ---- 128-byte K steps (conv_i8_r128): eligibility, launch
static bool r128_ok(const mhip_conv_i8_t *p) {
    const int C = p->in_c;
    return C >= 128 && (C & (C - 1)) == 0 && p->oc_pad % 128 == 0 && p->kh * p->kw <= 32 && (long)p->kh * p->kw * (p->kw - 1) < 65536 &&
           p->row_pad == p->kw * C && p->nseg <= 1 && in_extent_bytes(p) <= 0x7fffffffL && (long)p->oc_pad * p->kh * p->row_pad <= 0x7fffffffL;
}
// conv_i8_r128p: one persistent workgroup per CU (144 KB of LDS), aligned rows only (raw buffer stores)
static bool r128p_ok(const mhip_conv_i8_t *p) {
    return r128_ok(p) && p->safe && !p->add && !p->out_nchw && ((p->out_c | p->out_pix_stride | p->out_ch_off) & 15) == 0 &&
           persist_out_bytes(p) <= 0x7fffffffL;
}
template <bool HAS_LUT>
static int launch_r128p_t(const mhip_conv_i8_t *p, long total_pix) {
    const int k128 = p->kh * p->row_pad;
    const unsigned npt = (unsigned)((total_pix + 255) / 256), noc = (unsigned)(p->oc_pad / 128);
    const size_t lds = LUTB + 3 * (size_t)(256 + 128) * 128;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipFuncSetAttribute((const void *)conv_i8_r128p<HAS_LUT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return mhip_check(hipErrorUnknown, "conv_i8_r128p LDS attribute");
        cus = prop.multiProcessorCount;
    }
    unsigned ngrp = (unsigned)(tune().persist_slots > 0 ? tune().persist_slots : cus) / noc;
    if (ngrp < 1) ngrp = 1;
    if (ngrp > npt) ngrp = npt;
    int lg = 0;
    while ((1 << lg) < p->in_c) lg++;
    const unsigned magic = ((65536u + (unsigned)p->kw - 1u) / (unsigned)p->kw);
    hipLaunchKernelGGL((conv_i8_r128p<HAS_LUT>), dim3(noc * ngrp), dim3(512), lds, mhip_stream_native(), *p, (unsigned)total_pix, k128,
                       noc, npt, ngrp, lg, magic, make_fastdiv((unsigned)(p->out_h * p->out_w)), make_fastdiv((unsigned)p->out_w),
                       (unsigned)in_extent_bytes(p), (unsigned)persist_out_bytes(p));
    return mhip_check(hipGetLastError(), "conv_i8_r128p launch");
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
    if (form == 3) {
        if (!r128p_ok(p)) return -1;
        return p->lut ? launch_r128p_t<true>(p, total_pix) : launch_r128p_t<false>(p, total_pix);
    }
    return form == 2 ? launch_r128_t<256, 128, 8>(p, total_pix) : launch_r128_t<128, 128, 4>(p, total_pix);
}

// ---- two-team strip kernel (conv_i8_duo): geometry, eligibility, launch
struct duo_geom_t {
    duo_args_t a;
    size_t lds;
    unsigned grid;
};
static int device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus;
}
static bool duo_geom(const mhip_conv_i8_t *p, duo_geom_t *g) {
    const bool direct = !p->out_nchw && ((p->out_c | p->out_pix_stride | p->out_ch_off) & 15) == 0;
    const int C = p->in_c, hw = p->out_h * p->out_w, taps = p->kh * p->kw;
    if (!direct || !p->safe || (C != 64 && C != 128 && C != 256) || p->stride_w != 1 || p->stride_h != 1 || p->kh != 3 || p->kw != 3 || p->pad_top != 1 || p->pad_left != 1 || p->row_pad != p->kw * C || p->oc_pad % 128 != 0 || p->nseg > 1 || hw < 256 || p->add || persist_out_bytes(p) > 0x7fffffffL ||
        in_extent_bytes(p) > 0x7fffffffL)
        return false;
    const long total = (long)p->frames * hw;
    const int k64 = p->kh * p->row_pad; // = taps * C, a multiple of 64
    if (total <= 0 || total > 0x7fffffffL - 512 || (long)p->oc_pad * k64 > 0x7fffffffL || (k64 & 63)) return false;
    duo_args_t &a = g->a;
    memset(&a, 0, sizeof(a));
    a.k64 = k64;
    a.taps = taps;
    a.nchunk = C / 64;
    a.nks = taps * a.nchunk;
    a.cH = a.nchunk / 2;
    a.H = a.nchunk == 1 ? a.nks / 2 : a.cH * taps;
    a.PWP = (p->out_w - 1) * p->stride_w + p->kw;
    a.total_pix = (unsigned)total;
    a.ntiles = (unsigned)((total + 255) / 256);
    a.noc = (unsigned)(p->oc_pad / 128);
    // most patch rows any tile needs: tiles start at every multiple of 256 modulo the frame size
    int rows = 0;
    const unsigned cyc = a.ntiles < (unsigned)hw ? a.ntiles : (unsigned)hw;
    for (unsigned t = 0; t < cyc; t++) {
        const long g0 = (long)t * 256, gl = (g0 + 256 < total ? g0 + 256 : total) - 1;
        const long fA = g0 / hw, fB = gl / hw;
        const int oyA0 = (int)((g0 - fA * hw) / p->out_w), oyB1 = (int)((gl - fB * hw) / p->out_w);
        const int oyA1 = fA != fB ? p->out_h - 1 : oyB1;
        const int n = (oyA1 - oyA0) * p->stride_h + p->kh + (fA != fB ? oyB1 * p->stride_h + p->kh : 0);
        if (n > rows) rows = n;
    }
    // the last tile of the batch (cyc may not reach it)
    {
        const long g0 = (long)(a.ntiles - 1) * 256, gl = total - 1;
        const long fA = g0 / hw, fB = gl / hw;
        const int oyA0 = (int)((g0 - fA * hw) / p->out_w), oyB1 = (int)((gl - fB * hw) / p->out_w);
        const int oyA1 = fA != fB ? p->out_h - 1 : oyB1;
        const int n = (oyA1 - oyA0) * p->stride_h + p->kh + (fA != fB ? oyB1 * p->stride_h + p->kh : 0);
        if (n > rows) rows = n;
    }
    const long units = (long)rows * a.PWP * 4;
    a.ni = (int)((units + 511) / 512);
    if (a.ni < 1 || a.ni > taps - 1) return false; // the prefetch of an epoch must end two steps before the epoch does
    a.patch_bytes = a.ni * 8192;
    g->lds = LUTB + 128 * 4 + 512 + 3 * (size_t)128 * BK + 4 * (size_t)a.patch_bytes;
    if (g->lds > 160 * 1024) return false;
    a.dhw = make_fastdiv((unsigned)hw);
    a.dow = make_fastdiv((unsigned)p->out_w);
    a.dpwp = make_fastdiv((unsigned)a.PWP);
    a.out_bytes = (unsigned)persist_out_bytes(p);
    a.in_bytes = (unsigned)in_extent_bytes(p);
    a.kw_magic = (65536u + (unsigned)p->kw - 1u) / (unsigned)p->kw;
    unsigned ngrp = (unsigned)(tune().persist_slots > 0 ? tune().persist_slots : device_cus()) / a.noc;
    if (ngrp < 1) ngrp = 1;
    if (ngrp > (a.ntiles + 1) / 2) ngrp = (a.ntiles + 1) / 2;
    a.ngrp = ngrp;
    g->grid = a.noc * ngrp;
    return true;
}
template <bool HAS_LUT, bool HAS_ADD>
static int launch_duo_t(const mhip_conv_i8_t *p, const duo_geom_t &g) {
    static bool attr = false;
    if (!attr && hipFuncSetAttribute((const void *)conv_i8_duo<HAS_LUT, HAS_ADD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return mhip_check(hipErrorUnknown, "conv_i8_duo LDS attribute");
    attr = true;
    hipLaunchKernelGGL((conv_i8_duo<HAS_LUT, HAS_ADD>), dim3(g.grid), dim3(1024), g.lds, mhip_stream_native(), *p, g.a);
    return mhip_check(hipGetLastError(), "conv_i8_duo launch");
}
static int launch_duo(const mhip_conv_i8_t *p) {
    duo_geom_t g;
    if (!duo_geom(p, &g)) return -1;
    if (p->add) return p->lut ? launch_duo_t<true, true>(p, g) : launch_duo_t<false, true>(p, g);
    return p->lut ? launch_duo_t<true, false>(p, g) : launch_duo_t<false, false>(p, g);
}

static int launch_patch(const mhip_conv_i8_t *p, int k64, int th) {
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
static int pre_tile_rows(const mhip_conv_i8_t *p) {
    patch_geom_t g;
    for (int th : {16, 8, 4})
        if (patch_geom(p, th, &g) && (th == 4 || g.dbl)) return th;
    return 0;
}
extern "C" int mhip_conv_i8_pre_ok(const mhip_conv_i8_t *p) {
    if (!p || !p->pre_w || p->nseg > 1 || p->out_nchw) return 0;
    return pre_tile_rows(p) != 0;
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
    v.pws = 0;
    v.duo = 0;
    v.r128 = 0;
    // wide, shallow k x k layers: the patch-staged kernel wins wherever its double-buffered form fits (measured on the
    // 160x160 and 80x80 layers of yolov5s: 1.25-1.9x over the implicit-GEMM forms)
    patch_geom_t g;
    // Few frames: a launch whose large-batch tiling yields fewer workgroups than the device has CUs is bound by the
    // latency of ONE workgroup's K walk, not by bytes per MAC.  Smaller tiles put more CUs to work and shorten every
    // step: 4-row patches, the 128 x 128 tile with 128-byte K steps for deep K loops (half the barriers per MAC, four
    // waves), the plain two-stage tile walker without the up-front weight fetch for 1 x 1 layers.  These are the
    // autotuner's choices at batch 1 (yolov5s twin, 640 x 640: 0.62 -> 0.53 ms per frame).
    const long wg_large = ((long)p->frames * p->out_h * p->out_w + 255) / 256 * ((p->oc_pad + 127) / 128);
    const bool few = wg_large < 256 && tune().persist && !tune().bpx && !tune().stages && tune().small_batch;
    if (few) {
        if (patch_geom(p, 4, &g)) {
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
    if (tune().persist && !tune().bpx && !tune().stages) {
        // 16 output rows per workgroup wherever that patch fits at all (single-buffered included: the taller patch
        // re-reads fewer halo rows, measured 5-40 % over 8 rows on the 160x160 / 80x80 layers), else 8 rows
        if (patch_geom(p, 16, &g)) {
            v.persist = 0; v.bpx = 0; v.stages = 0; v.patch = 16;
            return v;
        }
        if (patch_geom(p, 8, &g) && g.dbl) {
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
    v.stages = v.persist ? tune().persist_stages : (tune().stages ? tune().stages : (nks <= 4 ? 2 : 3));
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
        return p->lut ? launch_persist_t<BPX, BN, 3, true>(p, total_pix, k64, lg, magic)
                      : launch_persist_t<BPX, BN, 3, false>(p, total_pix, k64, lg, magic);
    }
    if (v.w8) return BN == 128 ? launch_mfma<256, 128, 3, 1, 8>(p, total_pix, k64) : -1;
    if (v.ks2) return BPX == 128 && BN >= 64 ? launch_mfma<128, (BN >= 64 ? BN : 64), 2, 2>(p, total_pix, k64) : -1;
    return v.stages == 2 ? launch_mfma<BPX, BN, 2>(p, total_pix, k64) : launch_mfma<BPX, BN, 3>(p, total_pix, k64);
}

static int launch_variant(const mhip_conv_i8_t *p, long total_pix, int k64, const variant_t &v) {
    if (v.r128) return launch_r128(p, total_pix, v.r128);
    if (v.duo) return launch_duo(p);
    if (v.pws) return launch_pws(p, k64);
    if (v.patch) return launch_patch(p, k64, v.patch);
    const int bn = p->oc_pad % 128 == 0 ? 128 : (p->oc_pad % 64 == 0 ? 64 : 32);
    if (v.bpx == 256) {
        if (bn == 128) return launch_variant_t<256, 128>(p, total_pix, k64, v);
        if (bn == 64) return launch_variant_t<256, 64>(p, total_pix, k64, v);
        return launch_variant_t<256, 32>(p, total_pix, k64, v);
    }
    if (bn == 128) return launch_variant_t<128, 128>(p, total_pix, k64, v);
    if (bn == 64) return launch_variant_t<128, 64>(p, total_pix, k64, v);
    return launch_variant_t<128, 32>(p, total_pix, k64, v);
}

// Two convolutions over the same input in one launch (conv_i8_persist<PAIR>).  -2 = the pair is not eligible (the
// caller launches them one after the other), otherwise the launch's return code.
template <int BPX, int BN>
static int launch_pair_t(const mhip_conv_i8_t *a, const mhip_conv_i8_t *b, long total_pix, int k64, int lg, unsigned magic,
                         int wres) {
    if (a->nseg > 1)
        return launch_persist_t<BPX, BN, 2, true, true, true>(a, total_pix, k64, lg, magic, b, wres);
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
    if (bpx == 256) {
        if (bn == 128) return launch_pair_t<256, 128>(a, b, total_pix, k64, lg, magic, wres);
        if (bn == 64) return launch_pair_t<256, 64>(a, b, total_pix, k64, lg, magic, wres);
        return launch_pair_t<256, 32>(a, b, total_pix, k64, lg, magic, wres);
    }
    if (bn == 128) return launch_pair_t<128, 128>(a, b, total_pix, k64, lg, magic, wres);
    if (bn == 64) return launch_pair_t<128, 64>(a, b, total_pix, k64, lg, magic, wres);
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
        patch_geom_t g;
        const int first = pre_tile_rows(p);
        for (int th : {first, 16, 8, 4})
            if (th && n < max && patch_geom(p, th, &g)) {
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
        patch_geom_t g;
        if (v.patch && !patch_geom(p, v.patch, &g)) continue;
        if (v.persist && !persist_eligible(p)) continue;
        if (v.ks2 && ((nks & 1) || nks < 4 || p->oc_pad % 64 != 0)) continue;
        if (v.w8 && (p->oc_pad % 128 != 0 || nks < 3)) continue;
        pws_geom_t pg;
        if (v.pws && !pws_geom(p, &pg)) continue;
        duo_geom_t dg;
        if (v.duo && !duo_geom(p, &dg)) continue;
        if (v.r128 && !(v.r128 == 3 ? r128p_ok(p) : r128_ok(p))) continue;
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
    if (mhip_conv_i8_small_c(p->in_c, p->kw, p->out_c)) {
        const int PH = (SC_TH - 1) * p->stride_h + p->kh, PW = (SC_TW - 1) * p->stride_w + p->kw;
        if (p->stride_h >= 1 && p->stride_w >= 1 && total_pix <= 0x7fffffffL) {
            const int rc = oc_pad == 32 ? try_rgb<2>(p, k64) : try_rgb<4>(p, k64);
            if (rc != -2) return rc;
        }
        {
            const int PWp = (PW + 8 + 3) & ~3;
            const bool direct = !p->out_nchw && ((p->out_c | p->out_pix_stride | p->out_ch_off) & 15) == 0;
            const size_t lds = (size_t)oc_pad * k64 + 2 * ((((size_t)PH + 1) * PWp * 4 + 15) & ~(size_t)15) +
                               (direct ? 0 : (size_t)SC_BP * (oc_pad + OPAD)) + LUTB + (size_t)SC_BP * 8 + (size_t)oc_pad * 4;
            const long ntiles = (long)((p->out_w + SC_TW - 1) / SC_TW) * ((p->out_h + SC_TH - 1) / SC_TH) * p->frames;
            if (p->stride_h >= 1 && p->stride_w >= 1 && (long)PH * ((PW + 3) / 4) <= 2 * NTHREADS && total_pix <= 0x7fffffffL &&
                lds <= 64 * 1024 && ntiles < 0x0fffffffL)
                return oc_pad == 32 ? launch_smallc<2>(p, k64) : launch_smallc<4>(p, k64);
        }
        // a patch larger than the small-channel kernel stages (large strides / kernels): the gather kernel reads the same packing
        if (p->add || p->nseg > 1) return -1;
        if (oc_pad % 64 == 0) return launch_generic<64>(p, total_pix, k64);
        return launch_generic<32>(p, total_pix, k64);
    }
    if (p->pre_w) { // fused bottleneck: the patch-staged kernel at the tallest tile that fits (4 rows for few workgroups)
        if (!mhip_zero_page() || !mhip_conv_i8_pre_ok(p)) return -1;
        int th = pre_tile_rows(p);
        const int code = p->variant ? p->variant : tune().variant;
        patch_geom_t g;
        if (code >= 9 && code <= 11) { // forced (autotuner / tests): that height if it fits
            const int want = code == 10 ? 16 : (code == 9 ? 8 : 4);
            if (patch_geom(p, want, &g)) th = want;
        } else if (tune().small_batch && patch_geom(p, 4, &g) &&
                   ((long)p->frames * p->out_h * p->out_w + 255) / 256 * ((p->oc_pad + 127) / 128) < 256) {
            th = 4;
        }
        return launch_patch(p, k64, th);
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
