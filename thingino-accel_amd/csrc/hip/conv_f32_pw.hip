// conv_f32_pw.hip -- float32 1 x 1 convolutions (NCHW / OIHW, stride 1, no padding; reference src/mars/mxu_conv.c:673-710) on the
// bf16 matrix cores with split operands (use_mfma == 3: three piece products; conv_f32_split.hip explains the arithmetic).  Round 5.
//
// 31 of the yolov5s float twin's 60 convolutions are 1 x 1 and they were 10.4 of its 23 ms per batch under conv_f32_split: one
// barrier per K step with every wave in the same phase, ~150 vector instructions of split / mask / address work per step beside 48
// MFMAs (4600 - 5000 cycles per step for 1536 of matrix work).  This kernel is conv_f32_patch's two-phase structure without a
// patch: a 1 x 1 needs no halo, so a tile is BN consecutive pixels of the stacked frames and a K step's input is 32 channel rows of
// BN contiguous floats.
//   R(t): the input of step t + 1 (loaded during R(t - 1): 4 pixels x CPI channels per thread) is split once into bf16 hi / mid and
//         written channels-last into the LDS slot (t + 1) & 1 ([4 channel groups][BN pixels][8 hi | 8 mid]); the weights of step
//         t + 1 go registers -> LDS; the fragments of step t are read; the loads of step t + 3 (input and weights) are issued.
//   M(t): the step's MFMAs, nothing else.  The two waves of a SIMD run one phase apart (waves 4-7 one barrier behind).
// LDS hazards: a slot / stage written in R(t) was last read in an R(t - 1), which for either group ended at the barrier before this
// R(t) begins; it is first read in an R(t + 1).  Weights: conv_f32_split's image (natural channel order, two planes).
// Persistent over pixel tiles (the K pipeline runs through tile boundaries).  Takes stride 1, pad 0, 1 x 1, map sizes that are
// multiples of 4 pixels; everything else stays with conv_f32_split.
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

#define PW_NT 512

struct pwdiv_t {
    unsigned m, s1, s2;
};
__device__ __forceinline__ unsigned pwdiv(unsigned n, const pwdiv_t d) {
    const unsigned q = __umulhi(d.m, n);
    return (q + ((n - q) >> d.s1)) >> d.s2;
}
static pwdiv_t make_pwdiv(unsigned d) {
    pwdiv_t r;
    unsigned l = 0;
    while ((1ull << l) < d) l++;
    r.m = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    r.s1 = l < 1 ? l : 1;
    r.s2 = l > 0 ? l - 1 : 0;
    return r;
}

struct pw_args_t {
    int C, nks, kp, oc_pad; // input channels, K steps (even; the weight rows are padded), weight row length, weight rows per plane
    unsigned hw, total_pix, ntiles, in_bytes, per;
    pwdiv_t dhw;
};

__device__ __forceinline__ int pw_lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 2)) << 4); }
__device__ __forceinline__ float pw_silu(float v) {
    const float e = __builtin_amdgcn_exp2f(v * -1.44269504088896341f);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}
template <int N>
__device__ __forceinline__ void pw_split(const float (&x)[N], int (&hi)[N / 2], int (&mid)[N / 2]) { // as conv_f32_patch.hip's psplit
#pragma unroll
    for (int i = 0; i < N / 2; i++) {
        const f32x2 v = {x[2 * i], x[2 * i + 1]};
        const int h = __builtin_bit_cast(int, __builtin_convertvector(v, bf16x2));
        const float h0 = __int_as_float(h << 16), h1 = __int_as_float(h & (int)0xffff0000);
        f32x2 r;
        asm("v_sub_f32 %0, %1, %2" : "=v"(r[0]) : "v"(v[0]), "v"(h0));
        asm("v_sub_f32 %0, %1, %2" : "=v"(r[1]) : "v"(v[1]), "v"(h1));
        r[0] = __builtin_isfinite(h0) ? r[0] : 0.0f;
        r[1] = __builtin_isfinite(h1) ? r[1] : 0.0f;
        hi[i] = h;
        mid[i] = __builtin_bit_cast(int, __builtin_convertvector(r, bf16x2));
    }
}

// BM output channels x BN pixels per workgroup; waves WM (channels) x WN (pixels)
template <int BM, int WM, int WN, int BN>
__global__ __launch_bounds__(PW_NT, 2) void conv_f32_pw(const mhip_conv_f32_t p, const pw_args_t g) {
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int APLANE = BM * 64, WSTAGE = 2 * APLANE;
    constexpr int AE = BM * 32 / PW_NT, ATPR = 32 / AE, AD = AE / 2;
    constexpr int NPG = BN / 4;             // 4-pixel groups of a tile
    constexpr int CPI = 32 * NPG / PW_NT;   // channels per thread and step: 4 (BN 256) | 8 (BN 512)
    constexpr int NCS = 32 / CPI;           // channel sub-groups
    constexpr int PP = BN * 32 + 32;        // pitch of a channel-group plane: + 32 bytes, so that the four planes start in four different
                                            // quarters of the 128-byte bank period (see the lane mapping below)
    constexpr int SLOT = 4 * PP;            // [4 channel groups][BN pixels][32 bytes]
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    int8_t *wst = lds;              // two weight stages
    int8_t *xs = lds + 2 * WSTAGE;  // two input slots

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv % WM, wn = wv / WM;
    const int fr = lane & 15, fc = lane >> 4;
    const bool late = wv >= 4;
    const int oc0 = (int)blockIdx.y * BM;
    const unsigned hw = g.hw, plane_bytes = g.hw * 4u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, 0, (int)g.in_bytes, 0x00020000);

    // ---- this thread's share of a step's input: pixels 4 ipg .. + 3 of the tile, channels ics * CPI .. + CPI - 1 of the step's 32
    // Lane mapping: the channel sub-group varies FASTEST.  A lane's four pixels are four consecutive 32-byte records, so for a given
    // pixel slot every lane of a wave writes at the same offset inside the 128-byte bank period -- with the pixel group fastest the
    // eight LDS writes of a step were 16-way bank conflicts (first form of this kernel: 6000 cycles per step).  With the sub-group
    // fastest, 8 consecutive lanes write the 8 different 8-byte pieces of (4 planes x 32-byte period quarters x 2 halves): 2-way.  The
    // loads stay coalesced: the lanes of one load instruction that share a channel cover 128 contiguous bytes.
    const int ics = tid % NCS, ipg = tid / NCS;
    int8_t *xdst = xs + ((ics * CPI) >> 3) * PP + 4 * ipg * 32 + ((ics * CPI) & 7) * 2; // record of pixel 4 ipg in its channel group
    v4i bregs[2][CPI]; // two sets: the loads of step t + 2 are in flight while step t + 1's are split (two steps of latency)
    unsigned xoff; // byte offset of (frame, channel 0, position) of the lane's 4 pixels of the tile being fetched; ~0 = none
    auto fetch_setup = [&](unsigned t) __attribute__((always_inline)) {
        const unsigned q = t * BN + 4u * (unsigned)ipg;
        const unsigned f = pwdiv(q < g.total_pix ? q : 0u, g.dhw), pos = (q < g.total_pix ? q : 0u) - f * hw;
        xoff = t < g.ntiles && q < g.total_pix ? f * (unsigned)p.in_stride + pos * 4u : 0xffffffffu;
    };
    auto fetch_x = [&](int ks, v4i (&breg)[CPI]) __attribute__((always_inline)) { // step ks of the tile fetch_setup named
        // (the channel goes into the PER-LANE offset: the sub-group differs inside a wave, and a scalar offset that is not
        // wave-uniform makes the compiler serialise the load over its distinct values)
        const int c0 = ks * 32 + ics * CPI;
        unsigned vo = xoff == 0xffffffffu ? 0xffffffffu : xoff + (unsigned)c0 * plane_bytes;
#pragma unroll
        for (int j = 0; j < CPI; j++) {
            breg[j] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(xrs, c0 + j < g.C ? vo : 0xffffffffu, 0, 0));
            vo = vo == 0xffffffffu ? vo : vo + plane_bytes; // (beyond the last channel: zeros, like the weights there)
        }
    };
    auto commit_x = [&](int slot, const v4i (&breg)[CPI]) __attribute__((always_inline)) {
        int8_t *d = xdst + slot * SLOT;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float x[CPI];
#pragma unroll
            for (int j = 0; j < CPI; j++) x[j] = __int_as_float(breg[j][i]);
            int hi[CPI / 2], mid[CPI / 2];
            pw_split<CPI>(x, hi, mid);
            if (CPI == 8) {
                *(v4i *)(d + i * 32) = (v4i){hi[0], hi[1 % (CPI / 2)], hi[2 % (CPI / 2)], hi[3 % (CPI / 2)]};
                *(v4i *)(d + i * 32 + 16) = (v4i){mid[0], mid[1 % (CPI / 2)], mid[2 % (CPI / 2)], mid[3 % (CPI / 2)]};
            } else {
                *(int2 *)(d + i * 32) = make_int2(hi[0], hi[1 % (CPI / 2)]);
                *(int2 *)(d + i * 32 + 16) = make_int2(mid[0], mid[1 % (CPI / 2)]);
            }
        }
    };

    // ---- weights (conv_f32_split's image and register path)
    const int arow = tid / ATPR, akc = (tid % ATPR) * AE;
    const int8_t *wrow = (const int8_t *)p.w_split + ((size_t)(oc0 + arow) * g.kp + akc) * 2;
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
        const int aoff = pw_lds_off(arow, akc >> 3) + (akc & 7) * 2;
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            if (AE == 8) *(v4i *)(st + pl * APLANE + aoff) = (v4i){areg[pl][0], areg[pl][1 % AD], areg[pl][2 % AD], areg[pl][3 % AD]};
            else if (AE == 4) *(int2 *)(st + pl * APLANE + aoff) = make_int2(areg[pl][0], areg[pl][1 % AD]);
            else *(int *)(st + pl * APLANE + aoff) = areg[pl][0];
        }
    };

    // ---- A rows (pixels) of this lane: the same LDS addresses for every tile
    const int8_t *pbase[NI];
#pragma unroll
    for (int n = 0; n < NI; n++) pbase[n] = xs + fc * PP + (wn * TN + n * 16 + fr) * 32;
    unsigned ooff[NI];
    auto tile_setup = [&](unsigned t) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < NI; n++) {
            const unsigned q4 = t * BN + (unsigned)(wn * TN + n * 16 + fc * 4);
            const unsigned f4 = pwdiv(q4 < g.total_pix ? q4 : 0u, g.dhw), pos = (q4 < g.total_pix ? q4 : 0u) - f4 * hw;
            ooff[n] = q4 < g.total_pix ? f4 * (unsigned)p.out_stride + pos * 4u : 0xffffffffu;
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
    bf16x8 xh[NI], xm[NI], wh[MI], wmid[MI];
    auto read_frags = [&](int slot, int buf) __attribute__((always_inline)) {
        const int8_t *ap = wst + buf * WSTAGE;
#pragma unroll
        for (int c = 0; c < NI; c++) {
            const int8_t *a = pbase[c] + slot * SLOT;
            xh[c] = __builtin_bit_cast(bf16x8, *(const v4i *)a);
            xm[c] = __builtin_bit_cast(bf16x8, *(const v4i *)(a + 16));
        }
#pragma unroll
        for (int a = 0; a < MI; a++) {
            const int o = pw_lds_off(wm * TM + a * 16 + fr, fc);
            wh[a] = __builtin_bit_cast(bf16x8, *(const v4i *)(ap + o));
            wmid[a] = __builtin_bit_cast(bf16x8, *(const v4i *)(ap + APLANE + o));
        }
    };
    auto phase_m = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < MI; a++)
#pragma unroll
            for (int c = 0; c < NI; c++) {
                acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm[c], wh[a], acc[a][c], 0, 0, 0);
                acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[c], wmid[a], acc[a][c], 0, 0, 0);
                acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[c], wh[a], acc[a][c], 0, 0, 0);
            }
    };
    auto barrier_lds = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    const unsigned t_first = blockIdx.x * g.per;
    const unsigned t_end = (blockIdx.x + 1) * g.per < g.ntiles ? (blockIdx.x + 1) * g.per : g.ntiles;
    const int nks = g.nks;
    // ---- fetch cursors: the K pipeline runs on through tile boundaries, so "the step after next" may belong to the next tile (or,
    // with two steps per tile, to the one after): the input cursor (tile, step) and the weight cursor (step) advance one step per fetch
    unsigned ftile = t_first;
    int fstep = 0, wstep = 0;
    auto next_x = [&](v4i (&breg)[CPI]) __attribute__((always_inline)) {
        fetch_x(fstep, breg);
        if (++fstep == nks) {
            fstep = 0;
            ftile++;
            fetch_setup(ftile < t_end ? ftile : g.ntiles); // (beyond the run: nothing valid, zeros)
        }
    };
    auto next_w = [&](int (&areg)[2][AD]) __attribute__((always_inline)) {
        fetch_w(wstep, areg);
        if (++wstep == nks) wstep = 0;
    };
    // ---- prologue: step 0's input and weights in LDS; steps 1 and 2 in registers
    fetch_setup(t_first);
    next_x(bregs[0]);
    commit_x(0, bregs[0]);
    next_x(bregs[1]);
    next_x(bregs[0]);
    next_w(aregs[0]);
    commit_w(0, aregs[0]);
    next_w(aregs[1]);
    next_w(aregs[0]);
    __syncthreads();
    if (late) __builtin_amdgcn_s_barrier();
    for (unsigned t = t_first; t < t_end; t++) {
        tile_setup(t);
        for (int ks = 0; ks < nks; ks += 2) {
            // ---- R(ks): step ks + 1's input -> slot 1 and weights -> stage 1, fragments of ks, loads for step ks + 3
            commit_x(1, bregs[1]);
            commit_w(1, aregs[1]);
            read_frags(0, 0);
            next_w(aregs[1]);
            next_x(bregs[1]);
            barrier_lds();
            phase_m();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- R(ks + 1): step ks + 2's input -> slot 0 and weights -> stage 0, fragments of ks + 1, loads for step ks + 4
            commit_x(0, bregs[0]);
            commit_w(0, aregs[0]);
            read_frags(1, 1);
            next_w(aregs[0]);
            next_x(bregs[0]);
            barrier_lds();
            phase_m();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- store (as conv_f32_patch)
        if (p.add) {
            v4f addv[MI][NI];
#pragma unroll
            for (int c = 0; c < NI; c++)
#pragma unroll
                for (int a = 0; a < MI; a++) {
                    const int oc = oc0 + wm * TM + a * 16 + fr;
                    const bool ok = ooff[c] != 0xffffffffu && oc < p.out_c;
                    addv[a][c] = *(const v4f *)((const char *)p.add + (ok ? (size_t)ooff[c] + (size_t)oc * hw * 4u : 0));
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
                            for (int j = 0; j < 4; j++) r[j] = pw_silu(r[j]);
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
                            for (int j = 0; j < 4; j++) r[j] = pw_silu(r[j]);
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
    if (!late) __builtin_amdgcn_s_barrier();
}

static unsigned long g_pw_launches = 0;
extern "C" unsigned long mhip_conv_f32_pw_launches(void) { return g_pw_launches; }

template <int BM, int WM, int WN, int BN>
static int launch_pw(const mhip_conv_f32_t *p, pw_args_t &g) {
    auto kern = conv_f32_pw<BM, WM, WN, BN>;
    const size_t ldsb = 2 * 2 * (size_t)BM * 64 + 2 * 4 * ((size_t)BN * 32 + 32);
    static int cus = 0;
    if (!cus) {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return mhip_check(hipErrorUnknown, "conv_f32_pw attribute");
        cus = prop.multiProcessorCount;
    }
    const long total = (long)g.total_pix;
    g.ntiles = (unsigned)((total + BN - 1) / BN);
    const unsigned noc = (unsigned)((p->out_c + BM - 1) / BM);
    if (noc > 65535u) return -2;
    int slots = 0;
    mhip_conv_i8_tune_get("persist_slots", &slots);
    unsigned gx = (unsigned)(slots > 0 ? slots : cus) / noc;
    if (gx < 1) gx = 1;
    if (gx > g.ntiles) gx = g.ntiles;
    g.per = (g.ntiles + gx - 1) / gx;
    gx = (g.ntiles + g.per - 1) / g.per;
    hipLaunchKernelGGL(kern, dim3(gx, noc), dim3(PW_NT), ldsb, mhip_stream_native(), *p, g);
    g_pw_launches++;
    return mhip_check(hipGetLastError(), "conv_f32_pw");
}

// -2: not a shape this kernel takes (the caller goes on to conv_f32_split), else the launch result
int conv_f32_try_pw(const mhip_conv_f32_t *p) {
    if (!p->w_split || p->use_mfma != 3) return -2;
    if (p->kh != 1 || p->kw != 1 || p->stride_h != 1 || p->stride_w != 1 || p->pad_top || p->pad_left) return -2;
    if (p->in_h != p->out_h || p->in_w != p->out_w) return -2;
    const long hw = (long)p->out_h * p->out_w, total = hw * p->frames;
    if ((hw & 3) || p->in_c < 32) return -2; // (4-pixel groups stay inside a frame; fewer than one K step of channels: not worth a phase pair)
    const size_t in_bytes = (size_t)(p->frames - 1) * p->in_stride + (size_t)p->in_c * hw * 4;
    if (total > 0x7fffffffL - 512 || in_bytes > 0xfffffff0ull || (size_t)p->frames * p->out_stride > 0xfffffff0ull) return -2;
    if (p->add && p->add_stride != p->out_stride) return -2;
    pw_args_t g;
    memset(&g, 0, sizeof(g));
    g.C = p->in_c;
    g.kp = (int)((p->in_c + 63) / 64 * 64) + 64; // mhip_conv_f32_split_pack's row length for a 1 x 1 (K = in_c)
    g.nks = (g.kp - 64) / 32;
    g.oc_pad = (p->out_c + 127) / 128 * 128;
    g.hw = (unsigned)hw; g.total_pix = (unsigned)total; g.in_bytes = (unsigned)in_bytes;
    g.dhw = make_pwdiv((unsigned)hw);
    if (p->out_c > 64) return launch_pw<128, 2, 4, 256>(p, g);
    if (p->out_c > 32) return launch_pw<64, 1, 8, 256>(p, g); // (512-pixel tiles: 8 channels per thread and step, two sets of them: spills)
    return launch_pw<32, 1, 8, 512>(p, g);
}
