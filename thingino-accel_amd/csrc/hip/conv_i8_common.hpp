// conv_i8_common.hpp -- device helpers shared by the int8 convolution translation units (conv_i8.hip,
// conv_i8_patch.hip, conv_i8_stem.hip): LDS-DMA wrappers, the requantisation / LUT epilogue (reference
// src/mars/mxu_conv.c:722-754 restated for the VALU floor), exact division, XCD-aware block order, and the
// host-side hooks the translation units call across each other.  Design notes: DESIGN.md section 5.
#pragma once
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

// ---- launch anatomy (diagnostic build only: tools/anatomy_build.sh -> lib/diag/lib_anatomy.so, read by tools/launch_anatomy.py; round 6,
// VERDICT r5 item 4).  Thread 0 of every workgroup of the int8 convolution kernels stamps the 100 MHz constant clock (s_memrealtime: the
// same time base on every CU and XCD) at its start, when its prologue is over (tables / LUT / resident weights / first pipeline stages
// issued: the first counted wait of the K stream comes next), when its K stream is over (the last epilogue, or for a tile walker its last
// tile's, comes next) and at its end, and appends {launch key = output pointer, workgroup, hardware slot, 4 stamps} to a side buffer.
// The shipped library compiles none of it.
#ifdef ANATOMY
struct anat_rec_t {
    unsigned key, wg, hwid, xcc;
    unsigned long long t[4];
};
static __device__ anat_rec_t *g_anat_buf; // record 0 is the header: .key = records appended so far
static __device__ unsigned g_anat_cap;
#define ANAT_SETTER(NAME)                                                                                                    \
    extern "C" int NAME(void *buf, unsigned cap) {                                                                          \
        return hipMemcpyToSymbol(HIP_SYMBOL(g_anat_buf), &buf, sizeof buf) == hipSuccess &&                                  \
                       hipMemcpyToSymbol(HIP_SYMBOL(g_anat_cap), &cap, sizeof cap) == hipSuccess                             \
                   ? 0                                                                                                       \
                   : -1;                                                                                                     \
    }
#define ANAT_NOW(i)                                                                                                          \
    do {                                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        an_t[i] = __builtin_amdgcn_s_memrealtime();                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
    } while (0)
#define ANAT_BEGIN()                                                                                                         \
    unsigned long long an_t[4] = {0, 0, 0, 0};                                                                               \
    ANAT_NOW(0)
#define ANAT_END(P)                                                                                                          \
    do {                                                                                                                     \
        ANAT_NOW(3);                                                                                                         \
        if (threadIdx.x == 0 && g_anat_buf) {                                                                                \
            const unsigned an_i = atomicAdd(&g_anat_buf[0].key, 1u) + 1u;                                                    \
            if (an_i < g_anat_cap) {                                                                                         \
                anat_rec_t an_r;                                                                                             \
                an_r.key = (unsigned)((size_t)(P).out >> 4) ^ (unsigned)(P).out_ch_off;                                      \
                an_r.wg = blockIdx.x + gridDim.x * blockIdx.y;                                                               \
                an_r.hwid = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));  /* HW_ID: wave, simd, cu, sh, se */      \
                an_r.xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   /* XCC_ID */                              \
                for (int q = 0; q < 4; q++) an_r.t[q] = an_t[q];                                                             \
                g_anat_buf[an_i] = an_r;                                                                                     \
            }                                                                                                                \
        }                                                                                                                    \
    } while (0)
#else
#define ANAT_SETTER(NAME)
#define ANAT_NOW(i) do { } while (0)
#define ANAT_BEGIN() do { } while (0)
#define ANAT_END(P) do { } while (0)
#endif

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
// The compiler does not count LDS loads issued from asm: wait for them, and thread the loaded values through the wait so that no use is
// scheduled above it.  NV = 4 (one channel subtile: conv_i8_smallc<1>), 8 or 16 values per lane.
template <int NV>
__device__ __forceinline__ void wait_lds_values(int (&v)[NV]) {
    static_assert(NV == 4 || NV == 8 || NV == 16, "values per lane");
    if constexpr (NV == 16)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),
                       "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
    else if constexpr (NV == 8)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
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
// (Round 3, measured and dropped: ds_read_u8_d16 / ds_read_u8_d16_hi pairs sharing one register, which would make look-up and
// packing of four values 4 LDS + 1 vector instruction.  On this part a D16 load does NOT preserve the other half of its
// destination -- SRAM-ECC register files zero-fill it -- so the second load of a pair wipes the first: half of all bytes
// came out 0.  Merging the halves again costs the vector instructions the trick was meant to save.)
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
    wait_lds_values<NV>(v);
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
        wait_lds_values<NV>(v);
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
        wait_lds_values<NV>(v);
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
//  DIRECT (NHWC, 16-byte aligned rows): one 16-byte (WOC=4) / 8-byte (WOC=2) / 4-byte (WOC=1) global store per pixel straight
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
                else if (WOC == 2) { const uint2 t2 = *(const uint2 *)x; xw[0] = t2.x; xw[WOC > 1 ? 1 : 0] = t2.y; }
                else xw[0] = *(const uint32_t *)x;
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
                else if (WOC == 2) *(uint2 *)d = make_uint2(pk[0], pk[WOC > 1 ? 1 : 0]);
                else *(uint32_t *)d = pk[0]; // (WOC == 1: 16 channels per tile -- the small-channel stem of the shipped yolov5n files)
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
// host side: what the translation units of the int8 convolution share

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
    int few_wgs;        // MARS_HIP_FEW_WGS       a launch whose large-batch tiling gives fewer workgroups than this takes the small-tile policy
    int rows;           // MARS_HIP_ROWS          1: the default policy may pick conv_i8_rows (variant 20) where it measured faster
    int patch_ring;     // MARS_HIP_PATCH_RING    0: auto, else at most this many patch buffers per workgroup of the patch-staged kernel (1..4)
    int patch_lds_kb;   // MARS_HIP_PATCH_LDS_KB  LDS budget of one patch-staged workgroup (default 80: two workgroups per CU)
};
const tune_t &conv_i8_tune_state(); // conv_i8.hip

extern "C" int mhip_conv_i8_small_c(int in_c, int kw, int out_c);

// bytes from p->in to the end of the last frame's pixels (the buffer resource's range)
static inline long in_extent_bytes(const mhip_conv_i8_t *p) {
    return (long)(p->frames - 1) * (long)p->in_stride + (long)p->in_h * p->in_w * p->in_c;
}
// bytes from p->out to the end of the last pixel row the layer can write
static inline long persist_out_bytes(const mhip_conv_i8_t *p) {
    const long pstride = p->out_pix_stride ? p->out_pix_stride : p->out_c;
    return (long)(p->frames - 1) * (long)p->out_stride + (long)p->out_h * p->out_w * pstride;
}
static inline bool conv_i8_direct_rows(const mhip_conv_i8_t *p) { // NHWC rows stored straight from registers (as epilogue())
    return !p->out_nchw && ((p->out_c | p->out_pix_stride | p->out_ch_off) & 15) == 0;
}

// conv_i8_patch.hip: can the patch-staged kernel take this layer with `th` (4 | 8 | 16) tile rows?  *ring = patch buffers
// per workgroup it would run with (1 = the next tile's patch is not prefetched)
bool conv_i8_patch_ok(const mhip_conv_i8_t *p, int th, int *ring);
int conv_i8_launch_patch(const mhip_conv_i8_t *p, int k64, int th); // -1: not eligible
int conv_i8_pre_tile_rows(const mhip_conv_i8_t *p);                 // fused bottleneck: tallest tile that fits, 0 = none
// conv_i8_rows.hip: whole-row tiles, patch-staged input, streamed weights, one persistent workgroup per CU (variant 20)
bool conv_i8_rows_ok(const mhip_conv_i8_t *p);
int conv_i8_launch_rows(const mhip_conv_i8_t *p); // -1: not eligible
// conv_i8_stem.hip: -2 = not a shape that kernel takes, else the launch result
int conv_i8_try_rgb(const mhip_conv_i8_t *p, int k64);
int conv_i8_try_smallc(const mhip_conv_i8_t *p, int k64);
