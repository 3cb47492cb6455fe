// conv_i8.hip -- int8 convolution for gfx950 (MI355X): im2col-free implicit
// GEMM on v_mfma_i32_16x16x64_i8 with LDS-staged input and weight tiles and
// an in-register requantise / ReLU / LUT epilogue.
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
//   MFMA B operand = pixels   (N = 16 output pixels, all frames of the batch flattened)
//   K = kernel rows x (kw*in_c bytes, padded to 16): for NHWC the kw*in_c bytes
//       of one kernel row are CONTIGUOUS in the input, so a 16-byte K chunk is
//       one 16-byte global load -- no im2col buffer anywhere.
// Each lane ends up with 4 consecutive output channels of one pixel per
// accumulator, i.e. one packed dword store for NHWC output.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../mhip.h"

extern "C" hipStream_t mhip_stream_native(void);
extern "C" int mhip_check(hipError_t e, const char *what);

typedef int v4i __attribute__((ext_vector_type(4)));

#define BP 128      // pixels per workgroup
#define BK 64       // K bytes per step = one MFMA
#define NTHREADS 256

// LDS tile row = 64 bytes (4 chunks of 16).  XOR the chunk index with
// ((row>>2)&1)<<1: conflict-free for the ds_read_b128 lane groups of gfx950
// (MI355X_MICROARCH.md, LDS table) when 16 lanes read 16 consecutive rows.
__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * BK + (((chunk ^ ((row >> 1) & 2))) << 4);
}

__device__ __forceinline__ int requant(int acc, float cs) {
    float scaled = (float)acc * cs;
    float biased = scaled + (scaled >= 0.0f ? 0.5f : -0.5f);
    int r = (int)biased;                       // v_cvt_i32_f32: saturates, NaN -> 0
    if (!(biased < 2147483648.0f)) r = INT_MIN; // x86 cvttss2si: +overflow and NaN -> INT_MIN
    r = r > 127 ? 127 : r;
    r = r < -128 ? -128 : r;
    return r;
}

template <int BN, bool FAST>
__global__ __launch_bounds__(NTHREADS) void conv_i8_kernel(const mhip_conv_i8_t p, const long total_pix,
                                                           const int k64) {
    __shared__ __attribute__((aligned(16))) int8_t lds[2 * (BP + BN) * BK];
    constexpr int STAGE = (BP + BN) * BK; // one pipeline stage: X tile then W tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const long pix0 = (long)blockIdx.x * BP;
    const int oc0 = blockIdx.y * BN;
    const int hw = p.out_h * p.out_w;

    // ---- per-thread staging assignment: 2 pixel rows x one 16-byte chunk
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
    // position of this thread's chunk inside K: kernel row ky, tap kx, byte rc in tap (FAST)
    // or kernel row ky, byte r in the padded row (generic)
    int ky = 0, kx = 0, rc = cc * 16;
    if (FAST) {
        while (rc >= p.in_c) { rc -= p.in_c; kx++; }
        while (kx >= p.kw) { kx -= p.kw; ky++; }
    } else {
        while (rc >= p.row_pad) { rc -= p.row_pad; ky++; }
    }
    const int row_bytes = p.kw * p.in_c;

    const int nks = k64 / BK;
    v4i xreg[2];
    v4i wreg[(BN * 4 + NTHREADS - 1) / NTHREADS];
    constexpr int WLOADS = (BN * 4 + NTHREADS - 1) / NTHREADS;

    auto load_global = [&](int ks) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            v4i v = {0, 0, 0, 0};
            if (rvalid[j] && ky < p.kh) {
                int iy = iy0[j] + ky;
                if (iy >= 0 && iy < p.in_h) {
                    if (FAST) {
                        int ix = ix0[j] + kx;
                        if (ix >= 0 && ix < p.in_w)
                            v = *(const v4i *)(xbase[j] + ((size_t)iy * p.in_w + ix) * p.in_c + rc);
                    } else {
                        const int8_t *rowp = xbase[j] + ((long)iy * p.in_w + ix0[j]) * p.in_c;
                        int8_t b[16];
#pragma unroll
                        for (int e = 0; e < 16; e++) {
                            int rr = rc + e;
                            int8_t val = 0;
                            if (rr < row_bytes) {
                                int t = rr / p.in_c;
                                int ix = ix0[j] + t;
                                if (ix >= 0 && ix < p.in_w) val = rowp[rr];
                            }
                            b[e] = val;
                        }
                        v = *(v4i *)b;
                    }
                }
            }
            xreg[j] = v;
        }
#pragma unroll
        for (int j = 0; j < WLOADS; j++) {
            int idx = tid + j * NTHREADS; // chunk index in the BN x 4 tile
            if (idx < BN * 4) {
                int row = idx >> 2;
                wreg[j] = *(const v4i *)(p.w + (size_t)(oc0 + row) * k64 + ks * BK + (idx & 3) * 16);
            }
        }
        // advance this thread's K position by one step (64 bytes)
        rc += BK;
        if (FAST) {
            while (rc >= p.in_c) { rc -= p.in_c; kx++; }
            while (kx >= p.kw) { kx -= p.kw; ky++; }
        } else {
            while (rc >= p.row_pad) { rc -= p.row_pad; ky++; }
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            int row = (tid >> 2) + j * 64;
            *(v4i *)(lds + buf * STAGE + lds_off(row, cc)) = xreg[j];
        }
#pragma unroll
        for (int j = 0; j < WLOADS; j++) {
            int idx = tid + j * NTHREADS;
            if (idx < BN * 4) *(v4i *)(lds + buf * STAGE + BP * BK + lds_off(idx >> 2, idx & 3)) = wreg[j];
        }
    };

    constexpr int NS = BN / 16; // oc subtiles per wave
    v4i acc[NS][2];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        acc[s][0] = (v4i){0, 0, 0, 0};
        acc[s][1] = (v4i){0, 0, 0, 0};
    }

    load_global(0);
    store_lds(0);
    __syncthreads();

    const int frow = lane & 15, fchunk = lane >> 4;
    for (int ks = 0; ks < nks; ks++) {
        const int buf = ks & 1;
        if (ks + 1 < nks) load_global(ks + 1);
        v4i xb[2];
#pragma unroll
        for (int ps = 0; ps < 2; ps++) xb[ps] = *(const v4i *)(lds + buf * STAGE + lds_off(wv * 32 + ps * 16 + frow, fchunk));
#pragma unroll
        for (int s = 0; s < NS; s++) {
            v4i wa = *(const v4i *)(lds + buf * STAGE + BP * BK + lds_off(s * 16 + frow, fchunk));
            acc[s][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, xb[0], acc[s][0], 0, 0, 0);
            acc[s][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, xb[1], acc[s][1], 0, 0, 0);
        }
        if (ks + 1 < nks) store_lds(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds D[oc = s*16 + (lane>>4)*4 + r][pixel = lane&15]
    const bool vec4 = (p.out_c & 3) == 0 && !p.out_nchw;
#pragma unroll
    for (int ps = 0; ps < 2; ps++) {
        long pix = pix0 + wv * 32 + ps * 16 + (lane & 15);
        if (pix >= total_pix) continue;
        long f = pix / hw;
        int rem = (int)(pix - f * hw);
        int8_t *obase = p.out + (size_t)f * p.out_stride;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            int oc = oc0 + s * 16 + (lane >> 4) * 4;
            if (oc >= p.out_c) continue;
            int q[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int a = acc[s][ps][r];
                if (p.bias) a += p.bias[oc + r]; // bias is padded to oc_pad
                int v = requant(a, p.cs);
                if (p.relu) v = v < 0 ? 0 : v;
                if (p.lut) v = (int8_t)p.lut[v + 128];
                q[r] = v;
            }
            if (vec4) {
                uint32_t pk = (uint32_t)(q[0] & 255) | ((uint32_t)(q[1] & 255) << 8) |
                              ((uint32_t)(q[2] & 255) << 16) | ((uint32_t)(q[3] & 255) << 24);
                *(uint32_t *)(obase + (size_t)rem * p.out_c + oc) = pk;
            } else if (!p.out_nchw) {
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (oc + r < p.out_c) obase[(size_t)rem * p.out_c + oc + r] = (int8_t)q[r];
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (oc + r < p.out_c) obase[(size_t)(oc + r) * hw + rem] = (int8_t)q[r];
            }
        }
    }
}

extern "C" void mhip_conv_i8_pack_geom(int in_c, int kw, int out_c, int *row_pad, int *oc_pad) {
    if (row_pad) *row_pad = (kw * in_c + 15) & ~15;
    if (oc_pad) *oc_pad = (out_c + 31) & ~31;
}

template <int BN, bool FAST>
static int launch(const mhip_conv_i8_t *p, long total_pix, int k64) {
    dim3 grid((unsigned)((total_pix + BP - 1) / BP), (unsigned)(p->oc_pad / BN));
    hipLaunchKernelGGL((conv_i8_kernel<BN, FAST>), grid, dim3(NTHREADS), 0, mhip_stream_native(), *p, total_pix, k64);
    return mhip_check(hipGetLastError(), "conv_i8 launch");
}

extern "C" int mhip_conv_i8(const mhip_conv_i8_t *p) {
    // host-side shape checks: the kernel trusts these (a faulting kernel can reset the node)
    if (!p || !p->in || !p->out || !p->w) return -1;
    if (p->frames <= 0 || p->in_h <= 0 || p->in_w <= 0 || p->in_c <= 0 || p->out_h <= 0 || p->out_w <= 0 ||
        p->out_c <= 0 || p->kh <= 0 || p->kw <= 0 || p->stride_h <= 0 || p->stride_w <= 0)
        return -1;
    int row_pad, oc_pad;
    mhip_conv_i8_pack_geom(p->in_c, p->kw, p->out_c, &row_pad, &oc_pad);
    if (row_pad != p->row_pad || oc_pad != p->oc_pad) return -1;
    const long total_pix = (long)p->frames * p->out_h * p->out_w;
    const int k64 = (p->kh * p->row_pad + BK - 1) / BK * BK;
    const bool fast = (p->in_c % 16) == 0;
    if (total_pix <= 0 || (total_pix + BP - 1) / BP > 0x7fffffffL) return -1;
    if (oc_pad % 128 == 0) return fast ? launch<128, true>(p, total_pix, k64) : launch<128, false>(p, total_pix, k64);
    if (oc_pad % 64 == 0) return fast ? launch<64, true>(p, total_pix, k64) : launch<64, false>(p, total_pix, k64);
    return fast ? launch<32, true>(p, total_pix, k64) : launch<32, false>(p, total_pix, k64);
}
