// move.hip -- byte-moving layers of the .mars executor on gfx950: max-pool,
// channel concat, nearest up-sampling, and the NCHW->NHWC relayout that feeds
// the MFMA conv kernel when a graph is tagged NCHW.
//
// Index math is the reference's (src/mars/mars_runtime.c): all three layers
// read shape[1..3] as H, W, C and move int8 BYTES whatever the tensor's tag or
// dtype (:919-957 maxpool, :971-999 concat, :1014-1041 upsample); maxpool has
// no padding, clips its window at the bottom/right edge and starts from -128.
// All are HBM-bound; consecutive lanes touch consecutive channel bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../mhip.h"

extern "C" hipStream_t mhip_stream_native(void);
extern "C" int mhip_check(hipError_t e, const char *what);

typedef int v4i __attribute__((ext_vector_type(4)));
#define MV_THREADS 256

static inline dim3 mv_grid(size_t work, int frames) {
    size_t bx = (work + MV_THREADS - 1) / MV_THREADS;
    return dim3((unsigned)(bx ? bx : 1), (unsigned)frames);
}

// ----------------------------------------------------------------- maxpool
// VEC = channels handled per thread (4 when ch % 4 == 0 and pointers aligned)
template <int VEC>
__global__ __launch_bounds__(MV_THREADS) void maxpool_kernel(const int8_t *in, size_t is, int8_t *out, size_t os,
                                                             int in_h, int in_w, int ch, int out_h, int out_w, int kh,
                                                             int kw, int sh, int sw, int pstride, int choff) {
    const int cv = ch / VEC;
    size_t idx = (size_t)blockIdx.x * MV_THREADS + threadIdx.x;
    size_t total = (size_t)out_h * out_w * cv;
    if (idx >= total) return;
    int c = (int)(idx % cv) * VEC;
    size_t pix = idx / cv;
    int ox = (int)(pix % out_w), oy = (int)(pix / out_w);
    const int8_t *src = in + (size_t)blockIdx.y * is;
    int best[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) best[v] = -128;
    for (int ky = 0; ky < kh; ky++) {
        int iy = oy * sh + ky;
        if (iy >= in_h) break;
        for (int kx = 0; kx < kw; kx++) {
            int ix = ox * sw + kx;
            if (ix >= in_w) break;
            const int8_t *q = src + ((size_t)iy * in_w + ix) * ch + c;
            if (VEC == 4) {
                uint32_t w = *(const uint32_t *)q;
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    int e = (int8_t)(w >> (8 * v));
                    best[v] = e > best[v] ? e : best[v];
                }
            } else {
                int e = q[0];
                best[0] = e > best[0] ? e : best[0];
            }
        }
    }
    int8_t *dst = out + (size_t)blockIdx.y * os + pix * pstride + choff + c;
    if (VEC == 4) {
        *(uint32_t *)dst = (uint32_t)(best[0] & 255) | ((uint32_t)(best[1] & 255) << 8) |
                           ((uint32_t)(best[2] & 255) << 16) | ((uint32_t)(best[3] & 255) << 24);
    } else {
        dst[0] = (int8_t)best[0];
    }
}


// 16 channels per thread: 16-byte loads, bytes widened to packed int16 pairs so the running
// maximum is two v_pk_max_i16 per dword and tap.
typedef short s2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(MV_THREADS) void maxpool16_kernel(const int8_t *in, size_t is, int8_t *out, size_t os,
                                                               int in_h, int in_w, int ch, int out_h, int out_w, int kh,
                                                               int kw, int sh, int sw, int pstride, int choff) {
    const int cv = ch >> 4;
    size_t idx = (size_t)blockIdx.x * MV_THREADS + threadIdx.x;
    if (idx >= (size_t)out_h * out_w * cv) return;
    const int c = (int)(idx % cv) * 16;
    const size_t pix = idx / cv;
    const int ox = (int)(pix % out_w), oy = (int)(pix / out_w);
    const int8_t *src = in + (size_t)blockIdx.y * is;
    s2v ev[4], od[4];
    const s2v lowest = {-128, -128};
#pragma unroll
    for (int d = 0; d < 4; d++) ev[d] = od[d] = lowest;
    for (int ky = 0; ky < kh; ky++) {
        const int iy = oy * sh + ky;
        if (iy >= in_h) break;
        for (int kx = 0; kx < kw; kx++) {
            const int ix = ox * sw + kx;
            if (ix >= in_w) break;
            const v4i v = *(const v4i *)(src + ((size_t)iy * in_w + ix) * ch + c);
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const int w = v[d];
                const s2v x = *(const s2v *)&w;
                const s2v e = (s2v)(x << (short)8) >> (short)8; // sign-extended even bytes
                const s2v o = x >> (short)8;                    // sign-extended odd bytes
                ev[d] = __builtin_elementwise_max(ev[d], e);
                od[d] = __builtin_elementwise_max(od[d], o);
            }
        }
    }
    v4i r;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const s2v m = (ev[d] & (short)0xFF) | (s2v)(od[d] << (short)8);
        r[d] = *(const int *)&m;
    }
    *(v4i *)(out + (size_t)blockIdx.y * os + pix * pstride + choff + c) = r;
}

// Stride-1 pools over rows of W x C bytes with C % 4 == 0 but not % 16 (the float twins' byte-wise MAXPOOL: a 20 x 20 x 256 float map
// read as H = 256, W = 20, C = 20 bytes -- mars_runtime.c:919-957 takes shape[1..3] whatever the dtype).  Separable: a thread owns one dword
// column of the flattened rows and MP_RPT output rows; per input row ONE pass over the kw taps (dwords ch bytes apart) gives the row's
// horizontal maximum, the vertical maximum is taken over those in registers -- kw loads per output dword instead of kh * kw
// (the generic kernel above: 25 dword loads per output dword, 0.12 ms per pool of config 5 = 440 GB/s).  Window: the reference's
// (anchored top-left, clipped at the right and bottom edge).
#define MP_RPT 8
#define MP_KMAX 8
__global__ __launch_bounds__(MV_THREADS) void maxpool_rows_kernel(const int8_t *in, size_t is, int8_t *out, size_t os, int in_h, int in_w,
                                                                  int ch, int out_h, int out_w, int kh, int kw) {
    const int cdw = out_w * ch / 4; // dword columns of an output row
    const size_t idx = (size_t)blockIdx.x * MV_THREADS + threadIdx.x;
    const int strips = (out_h + MP_RPT - 1) / MP_RPT;
    if (idx >= (size_t)cdw * strips) return;
    const int cd = (int)(idx % cdw), oy0 = (int)(idx / cdw) * MP_RPT;
    const int b = cd * 4, ox = b / ch;
    const int8_t *src = in + (size_t)blockIdx.y * is;
    const int rowb = in_w * ch;
    int ntap = in_w - ox; // taps inside the row
    ntap = ntap < kw ? ntap : kw;
    const s2v lowest = {-128, -128};
    s2v hev[MP_RPT + MP_KMAX - 1], hod[MP_RPT + MP_KMAX - 1]; // horizontal maxima of input rows oy0 .. oy0 + MP_RPT + kh - 2 (even / odd bytes)
#pragma unroll
    for (int r = 0; r < MP_RPT + MP_KMAX - 1; r++) {
        s2v e = lowest, o = lowest;
        const int iy = oy0 + r;
        if (r < MP_RPT + kh - 1 && iy < in_h) {
            const int8_t *q = src + (size_t)iy * rowb + b;
            for (int kx = 0; kx < ntap; kx++) {
                const int w = *(const int *)(q + kx * ch);
                const s2v x = *(const s2v *)&w;
                e = __builtin_elementwise_max(e, (s2v)((s2v)(x << (short)8) >> (short)8));
                o = __builtin_elementwise_max(o, (s2v)(x >> (short)8));
            }
        }
        hev[r] = e; hod[r] = o;
    }
#pragma unroll
    for (int r = 0; r < MP_RPT; r++) {
        const int oy = oy0 + r;
        if (oy >= out_h) break;
        s2v e = lowest, o = lowest;
#pragma unroll
        for (int ky = 0; ky < MP_KMAX; ky++)
            if (ky < kh) { // (rows past the bottom edge hold `lowest`)
                e = __builtin_elementwise_max(e, hev[r + ky]);
                o = __builtin_elementwise_max(o, hod[r + ky]);
            }
        const s2v m = (e & (short)0xFF) | (s2v)(o << (short)8);
        *(int *)(out + (size_t)blockIdx.y * os + (size_t)oy * out_w * ch + b) = *(const int *)&m;
    }
}

extern "C" int mhip_maxpool_i8(const int8_t *in, size_t in_stride, int8_t *out, size_t out_stride, int frames,
                               int in_h, int in_w, int ch, int out_h, int out_w, int kh, int kw, int sh, int sw,
                               int out_pix_stride, int out_ch_off) {
    if (!in || !out || frames <= 0 || in_h < 0 || in_w < 0 || ch < 0 || out_h < 0 || out_w < 0 || kh < 0 || kw < 0 ||
        sh < 0 || sw < 0 || out_pix_stride < 0 || out_ch_off < 0)
        return -1;
    if (out_pix_stride && out_pix_stride < out_ch_off + ch) return -1;
    const int pstride = out_pix_stride ? out_pix_stride : ch, choff = out_ch_off;
    size_t total = (size_t)out_h * out_w * ch;
    if (total == 0) return 0;
    bool v16 = (ch % 16 == 0) && ((((uintptr_t)in | (uintptr_t)out | in_stride | out_stride | pstride | choff) & 15) == 0);
    bool v4 = (ch % 4 == 0) && ((((uintptr_t)in | (uintptr_t)out | in_stride | out_stride | pstride | choff) & 3) == 0);
    if (!v16 && v4 && sh == 1 && sw == 1 && pstride == ch && choff == 0 && kh >= 1 && kh <= MP_KMAX && kw >= 1 && out_w <= in_w && out_h <= in_h)
        hipLaunchKernelGGL(maxpool_rows_kernel, mv_grid((size_t)(out_w * ch / 4) * ((out_h + MP_RPT - 1) / MP_RPT), frames), dim3(MV_THREADS), 0,
                           mhip_stream_native(), in, in_stride, out, out_stride, in_h, in_w, ch, out_h, out_w, kh, kw);
    else if (v16)
        hipLaunchKernelGGL(maxpool16_kernel, mv_grid(total / 16, frames), dim3(MV_THREADS), 0, mhip_stream_native(), in,
                           in_stride, out, out_stride, in_h, in_w, ch, out_h, out_w, kh, kw, sh, sw, pstride, choff);
    else if (v4)
        hipLaunchKernelGGL((maxpool_kernel<4>), mv_grid(total / 4, frames), dim3(MV_THREADS), 0, mhip_stream_native(),
                           in, in_stride, out, out_stride, in_h, in_w, ch, out_h, out_w, kh, kw, sh, sw, pstride, choff);
    else
        hipLaunchKernelGGL((maxpool_kernel<1>), mv_grid(total, frames), dim3(MV_THREADS), 0, mhip_stream_native(), in,
                           in_stride, out, out_stride, in_h, in_w, ch, out_h, out_w, kh, kw, sh, sw, pstride, choff);
    return mhip_check(hipGetLastError(), "maxpool");
}

// A chain of up to three stride-1 max-pools with the same window over a small feature map (SPPF: 5x5 three times
// on 20x20): one workgroup holds a whole frame of 16 channels in LDS, widened to packed int16 once, and evaluates
// every pool separably (row maximum, then column maximum -- exact for integers), writing each stage's result.
// Window semantics are the reference's (mars_runtime.c:908-960): anchored top-left, clipped at the right and
// bottom edge, no padding.  One launch and one read of the input instead of three of each.
typedef struct {
    int8_t *out[3];
    size_t stride[3];
} pool_chain_outs_t;
__global__ __launch_bounds__(MV_THREADS) void pool_chain_kernel(const int8_t *in, size_t is, pool_chain_outs_t outs, int n,
                                                                int H, int W, int ch, int kh, int kw, int frames) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pc_lds[];
    const int hw = H * W;
    // [8][hw] dwords (plane d of pixel px at d*hw + px: consecutive lanes -> consecutive banks): planes 0-3 the even
    // bytes, 4-7 the odd bytes of the 16 channels, widened to 16 bits for v_pk_max_i16
    s2v *A = (s2v *)pc_lds;
    s2v *T = A + (size_t)hw * 8;      // row maxima, same layout
    // workgroup -> (frame, channel group): consecutive workgroup ids go round-robin over the 8 XCDs, each with its own
    // L2.  All channel groups of a frame are given to ONE XCD (ids xcd, xcd + 8, ...), so the 16-byte pieces they
    // write into the same 128-byte lines merge in that L2 instead of leaving 8 L2s with partial lines each
    const int ncg = ch / 16, slot = blockIdx.x >> 3;
    const int f = (slot / ncg) * 8 + (blockIdx.x & 7), c = (slot % ncg) * 16;
    if (f >= frames) return;
    const int8_t *src = in + (size_t)f * is + c;
    for (int px = threadIdx.x; px < hw; px += MV_THREADS) {
        const v4i v = *(const v4i *)(src + (size_t)px * ch);
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int w = v[d];
            const s2v x = *(const s2v *)&w;
            A[d * hw + px] = (s2v)(x << (short)8) >> (short)8;
            A[(4 + d) * hw + px] = x >> (short)8;
        }
    }
    __syncthreads();
    for (int stage = 0; stage < n; stage++) {
        for (int px = threadIdx.x; px < hw; px += MV_THREADS) { // row pass
            const int y = px / W, x = px - y * W;
            const int xe = x + kw < W ? x + kw : W;
            s2v m[8];
#pragma unroll
            for (int d = 0; d < 8; d++) m[d] = A[d * hw + px];
            for (int xx = x + 1; xx < xe; xx++)
#pragma unroll
                for (int d = 0; d < 8; d++) m[d] = __builtin_elementwise_max(m[d], A[d * hw + y * W + xx]);
#pragma unroll
            for (int d = 0; d < 8; d++) T[d * hw + px] = m[d];
        }
        __syncthreads();
        int8_t *dst = outs.out[stage] + (size_t)f * outs.stride[stage] + c;
        for (int px = threadIdx.x; px < hw; px += MV_THREADS) { // column pass, result back into A and out to HBM
            const int y = px / W, x = px - y * W;
            const int ye = y + kh < H ? y + kh : H;
            s2v m[8];
#pragma unroll
            for (int d = 0; d < 8; d++) m[d] = T[d * hw + px];
            for (int yy = y + 1; yy < ye; yy++)
#pragma unroll
                for (int d = 0; d < 8; d++) m[d] = __builtin_elementwise_max(m[d], T[d * hw + yy * W + x]);
            v4i r;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                A[d * hw + px] = m[d];
                A[(4 + d) * hw + px] = m[4 + d];
                const s2v b = (m[d] & (short)0xFF) | (s2v)(m[4 + d] << (short)8);
                r[d] = *(const int *)&b;
            }
            *(v4i *)(dst + (size_t)px * ch) = r;
        }
        __syncthreads();
    }
}

extern "C" int mhip_pool_chain_i8(const int8_t *in, size_t in_stride, int8_t *const *outs, const size_t *out_strides, int n,
                                  int frames, int h, int w, int ch, int kh, int kw) {
    if (!in || !outs || !out_strides || n < 1 || n > 3 || frames <= 0 || frames > (1 << 20) || h <= 0 || w <= 0 || ch <= 0 || ch > 65536 ||
        (ch & 15) || kh <= 0 || kw <= 0)
        return -1;
    const size_t lds = (size_t)h * w * 8 * 4 * 2;
    if (lds > 60 * 1024 || ((uintptr_t)in | in_stride) & 15) return -1;
    pool_chain_outs_t o;
    for (int i = 0; i < 3; i++) {
        o.out[i] = i < n ? outs[i] : nullptr;
        o.stride[i] = i < n ? out_strides[i] : 0;
        if (i < n && (!outs[i] || (((uintptr_t)outs[i] | out_strides[i]) & 15))) return -1;
    }
    const unsigned groups = (unsigned)((frames + 7) / 8) * (unsigned)(ch / 16) * 8u;
    hipLaunchKernelGGL(pool_chain_kernel, dim3(groups), dim3(MV_THREADS), lds, mhip_stream_native(),
                       in, in_stride, o, n, h, w, ch, kh, kw, frames);
    return mhip_check(hipGetLastError(), "pool chain");
}

// ------------------------------------------------------- padded rows -> dense
// dst[r * width + k] = src[r * pitch + k], k < width: graph outputs kept at an aligned row pitch (pad_output_rows of
// the host) packed into the reference's dense bytes before they leave the device.  One thread per 16 source bytes.
__global__ __launch_bounds__(MV_THREADS) void unpad_rows_kernel(const int8_t *src, int8_t *dst, size_t rows, int width, int pitch) {
    const int cpr = (width + 15) / 16;
    const size_t idx = (size_t)blockIdx.x * MV_THREADS + threadIdx.x;
    if (idx >= rows * (size_t)cpr) return;
    const size_t r = idx / (size_t)cpr;
    const int c = (int)(idx - r * (size_t)cpr) * 16;
    const v4i v = *(const v4i *)(src + r * (size_t)pitch + c);
    int8_t *d = dst + r * (size_t)width + c;
    if (c + 16 <= width) {
        __builtin_memcpy(d, &v, 16); // unaligned dwordx4 store
    } else {
        const int8_t *b = (const int8_t *)&v;
        for (int k = 0; k < width - c; k++) d[k] = b[k];
    }
}

extern "C" int mhip_unpad_rows(const void *src, void *dst, size_t rows, int width, int pitch) {
    if (!src || !dst || width <= 0 || pitch < width || (pitch & 15) || ((uintptr_t)src & 15)) return -1;
    if (rows == 0) return 0;
    const size_t n = rows * (size_t)((width + 15) / 16), blocks = (n + MV_THREADS - 1) / MV_THREADS;
    if (blocks > 0x7fffffffu) return -1;
    hipLaunchKernelGGL(unpad_rows_kernel, dim3((unsigned)blocks), dim3(MV_THREADS), 0, mhip_stream_native(), (const int8_t *)src,
                       (int8_t *)dst, rows, width, pitch);
    return mhip_check(hipGetLastError(), "unpad rows");
}

// ------------------------------------------------------------ concat slice
// out[(pix)*out_c + ch_off + c] = in[pix*in_c + c], pix over out_h*out_w
template <int VEC>
__global__ __launch_bounds__(MV_THREADS) void concat_kernel(const int8_t *in, size_t is, int8_t *out, size_t os,
                                                            size_t npix, int in_c, int out_c, int ch_off) {
    const int cv = in_c / VEC;
    size_t idx = (size_t)blockIdx.x * MV_THREADS + threadIdx.x;
    if (idx >= npix * cv) return;
    int c = (int)(idx % cv) * VEC;
    size_t pix = idx / cv;
    const int8_t *s = in + (size_t)blockIdx.y * is + pix * in_c + c;
    int8_t *d = out + (size_t)blockIdx.y * os + pix * out_c + ch_off + c;
    if (VEC == 16) *(v4i *)d = *(const v4i *)s;
    else if (VEC == 8) *(uint2 *)d = *(const uint2 *)s;
    else if (VEC == 4) *(int *)d = *(const int *)s;
    else d[0] = s[0];
}

extern "C" int mhip_concat_slice(const int8_t *in, size_t in_stride, int8_t *out, size_t out_stride, int frames,
                                 int out_h, int out_w, int in_c, int out_c, int ch_off) {
    if (!in || !out || frames <= 0 || out_h < 0 || out_w < 0 || in_c < 0 || out_c < 0 || ch_off < 0) return -1;
    size_t npix = (size_t)out_h * out_w;
    if (npix == 0 || in_c == 0) return 0;
    bool v16 = (in_c % 16 == 0) && (out_c % 16 == 0) && (ch_off % 16 == 0) &&
               ((((uintptr_t)in | (uintptr_t)out | in_stride | out_stride) & 15) == 0);
    // (the float32 twins' "channels" are map rows of 20 / 40 / 80 / 160 BYTES -- the reference's concat is byte logic over
    // shape[3] whatever the dtype: dwords, not single bytes, for those)
    const bool v4 = (in_c % 4 == 0) && (out_c % 4 == 0) && (ch_off % 4 == 0) &&
                    ((((uintptr_t)in | (uintptr_t)out | in_stride | out_stride) & 3) == 0);
    const bool v8 = (in_c % 8 == 0) && (out_c % 8 == 0) && (ch_off % 8 == 0) && ((((uintptr_t)in | (uintptr_t)out | in_stride | out_stride) & 7) == 0);
    if (v16)
        hipLaunchKernelGGL((concat_kernel<16>), mv_grid(npix * (in_c / 16), frames), dim3(MV_THREADS), 0,
                           mhip_stream_native(), in, in_stride, out, out_stride, npix, in_c, out_c, ch_off);
    else if (v8) // (the twins' 40-byte rows)
        hipLaunchKernelGGL((concat_kernel<8>), mv_grid(npix * (in_c / 8), frames), dim3(MV_THREADS), 0,
                           mhip_stream_native(), in, in_stride, out, out_stride, npix, in_c, out_c, ch_off);
    else if (v4)
        hipLaunchKernelGGL((concat_kernel<4>), mv_grid(npix * (in_c / 4), frames), dim3(MV_THREADS), 0,
                           mhip_stream_native(), in, in_stride, out, out_stride, npix, in_c, out_c, ch_off);
    else
        hipLaunchKernelGGL((concat_kernel<1>), mv_grid(npix * in_c, frames), dim3(MV_THREADS), 0,
                           mhip_stream_native(), in, in_stride, out, out_stride, npix, in_c, out_c, ch_off);
    return mhip_check(hipGetLastError(), "concat");
}

// ------------------------------------------------ the reference's CONCAT on NCHW-tagged tensors, pixels x channels on both sides
// The reference's concat copies runs of shape[3] bytes whatever the tag (mars_runtime.c:971-999).  On [1, C, H, W] tensors of equal H, W
// (every C3 concat of the shipped NCHW-tagged files) that is, in flat byte terms,  out[f + n W] = in_n[f]  for f in [0, C_out H W), input
// after input: input n's bytes land n map rows further down, the LAST input wins wherever two overlap, and bytes of in_n beyond its own
// C_n H W are the zeros of its private slack -- a shift by (N - 1) rows of the last input, not a channel concatenation.  This kernel
// produces exactly those bytes when inputs and output are held pixels x channels on the device (mars_plan.c nhwc_internal): a thread
// owns 16 channels of one output pixel; logical element (c, h, w) is flat j = (c H + h) W + w, its writer n = min(j / W, N - 1), its
// source flat f = j - n W -> (c', h', w') of input n at [(h' W + w') C_n + c'].  Rows h >= N - 1 (all but the first N - 1 of a map) read
// 16 consecutive channels of ONE source pixel: a 16-byte load; the rest go byte by byte.
struct concatq_args_t {
    const int8_t *in[4];
    size_t in_stride[4];
    int in_c[4];
    int n;
};
__global__ __launch_bounds__(MV_THREADS) void concat_nchwq_kernel(const concatq_args_t a, int8_t *out, size_t os, int out_c, int H, int W, int rows) {
    const int cg = out_c / 16;
    const size_t idx = (size_t)blockIdx.x * MV_THREADS + threadIdx.x;
    if (idx >= (size_t)rows * W * cg) return; // (rows = H, or only the first rows of the map: the rest is never read -- mars_plan.c virtual_concat_q)
    const int c0 = (int)(idx % cg) * 16;
    const int pix = (int)(idx / cg), h = pix / W, w = pix - h * W;
    const int last = a.n - 1;
    int8_t *d = out + (size_t)blockIdx.y * os + (size_t)pix * out_c + c0;
    if (h >= last) { // f = (c H + h - last) W + w: channel c, row h - last of the last input, for every c of the group
        v4i v = {0, 0, 0, 0};
        if (c0 < a.in_c[last]) v = *(const v4i *)(a.in[last] + (size_t)blockIdx.y * a.in_stride[last] + ((size_t)(h - last) * W + w) * a.in_c[last] + c0);
        *(v4i *)d = v;
        return;
    }
    const int HW = H * W;
    uint32_t wd[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int j = ((c0 + e) * H + h) * W + w;
        const int n = j / W < last ? j / W : last;
        const int f = j - n * W;
        uint32_t b = 0;
        if (f < a.in_c[n] * HW) {
            const int cs = f / HW, r = f - cs * HW; // r = h' W + w'
            b = (uint8_t)a.in[n][(size_t)blockIdx.y * a.in_stride[n] + (size_t)r * a.in_c[n] + cs];
        }
        wd[e >> 2] |= b << (8 * (e & 3));
    }
    *(v4i *)d = (v4i){(int)wd[0], (int)wd[1], (int)wd[2], (int)wd[3]};
}

extern "C" int mhip_concat_nchwq(const int8_t *const *ins, const size_t *in_strides, const int *in_c, int n, int8_t *out, size_t out_stride,
                                 int frames, int out_c, int H, int W, int rows_only) {
    const int rows = rows_only > 0 && rows_only < H ? rows_only : H;
    if (!ins || !in_strides || !in_c || !out || n < 1 || n > 4 || frames <= 0 || out_c <= 0 || (out_c & 15) || H <= 0 || W <= 0) return -1;
    if ((long)out_c * H * W > 0x7fffffffL || ((((uintptr_t)out | out_stride) & 15) != 0)) return -1;
    concatq_args_t a;
    memset(&a, 0, sizeof a);
    a.n = n;
    for (int k = 0; k < n; k++) {
        if (!ins[k] || in_c[k] <= 0 || (in_c[k] & 15) || ((((uintptr_t)ins[k] | in_strides[k]) & 15) != 0)) return -1;
        a.in[k] = ins[k]; a.in_stride[k] = in_strides[k]; a.in_c[k] = in_c[k];
    }
    hipLaunchKernelGGL(concat_nchwq_kernel, mv_grid((size_t)rows * W * (out_c / 16), frames), dim3(MV_THREADS), 0, mhip_stream_native(), a, out, out_stride,
                       out_c, H, W, rows);
    return mhip_check(hipGetLastError(), "concat (NCHW-tagged, pixels x channels)");
}

// ------------------------------- the reference's UPSAMPLE / MAXPOOL on NCHW-tagged tensors, pixels x channels on both sides (round 6)
// Both index shape[1..3] as H, W, C whatever the tag (mars_runtime.c:919-957, 1014-1041): on a [1, C, H, W] tensor their "rows" are
// channels, their "columns" map rows, their "channels" runs of W bytes.  As for the concat above: the result is a fixed function of flat
// byte indices, evaluated here for operands held pixels x channels (mars_plan.c nhwc_internal).
// upsample: quirk output (oy < qoh, ox < qow, c' < qch) at flat j = (oy qow + ox) qch + c' takes quirk input (min(oy / sh, qih - 1),
// min(ox / sw, qiw - 1), c'); bytes of the output tensor beyond qoh qow qch are never written (zero).  A thread owns 16 channels of one
// output pixel, byte by byte (two small tensors per graph).
__global__ __launch_bounds__(MV_THREADS) void upsample_nchwq_kernel(const int8_t *in, size_t is, int Ci, int HWi, int8_t *out, size_t os, int Co, int Ho,
                                                                    int Wo, int qih, int qiw, int qch, int qoh, int qow, int sh, int sw) {
    const int cg = Co / 16;
    const size_t idx = (size_t)blockIdx.x * MV_THREADS + threadIdx.x;
    if (idx >= (size_t)Ho * Wo * cg) return;
    const int c0 = (int)(idx % cg) * 16, pix = (int)(idx / cg);
    const unsigned written = (unsigned)qoh * qow * qch, row = (unsigned)qow * qch; // (every tensor below 2^31 bytes: the launcher checks)
    const int8_t *s = in + (size_t)blockIdx.y * is;
    uint32_t wd[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const unsigned j = (unsigned)(c0 + e) * (unsigned)(Ho * Wo) + (unsigned)pix;
        uint32_t b = 0;
        if (j < written) {
            const unsigned oy = j / row, r = j - oy * row, ox = r / (unsigned)qch, cq = r - ox * (unsigned)qch;
            unsigned iy = oy / (unsigned)sh, ix = ox / (unsigned)sw;
            iy = iy > (unsigned)qih - 1 ? (unsigned)qih - 1 : iy;
            ix = ix > (unsigned)qiw - 1 ? (unsigned)qiw - 1 : ix;
            const unsigned f = (iy * (unsigned)qiw + ix) * (unsigned)qch + cq; // flat byte of the input = (ci Hi + hi) Wi + wi
            const unsigned ci = f / (unsigned)HWi, rr = f - ci * (unsigned)HWi;
            b = (uint8_t)s[(size_t)rr * Ci + ci];
        }
        wd[e >> 2] |= b << (8 * (e & 3));
    }
    *(v4i *)(out + (size_t)blockIdx.y * os + (size_t)pix * Co + c0) = (v4i){(int)wd[0], (int)wd[1], (int)wd[2], (int)wd[3]};
}

// The case every shipped graph has -- [1, C, H, W] -> [1, C, 2H, 2W], factors 2 / 2: in flat terms the reference's loop writes
// out(co, ho, wo) = in(co, ho mod H, wo mod W) for co < C / 2 (oy = 2 co + (ho >= H), ox = 2 (ho mod H) + (wo >= W), both halved again by
// the factors) and nothing for co >= C / 2: the map tiled 2 x 2 into the first half of the channels.  One 16-byte copy per thread.
__global__ __launch_bounds__(MV_THREADS) void upsample_nchwq_tile2_kernel(const int8_t *in, size_t is, int8_t *out, size_t os, int C, int H, int W) {
    const int cg = C / 16, Wo = 2 * W;
    const size_t idx = (size_t)blockIdx.x * MV_THREADS + threadIdx.x;
    if (idx >= (size_t)4 * H * W * cg) return;
    const int c0 = (int)(idx % cg) * 16, pix = (int)(idx / cg), ho = pix / Wo, wo = pix - ho * Wo;
    v4i v = {0, 0, 0, 0};
    if (c0 < C / 2) v = *(const v4i *)(in + (size_t)blockIdx.y * is + ((size_t)(ho >= H ? ho - H : ho) * W + (wo >= W ? wo - W : wo)) * C + c0);
    *(v4i *)(out + (size_t)blockIdx.y * os + (size_t)pix * C + c0) = v;
}

extern "C" int mhip_upsample_nchwq(const int8_t *in, size_t in_stride, int Ci, int Hi, int Wi, int8_t *out, size_t out_stride, int Co, int Ho, int Wo,
                                   int frames, int qih, int qiw, int qch, int qoh, int qow, int sh, int sw) {
    if (!in || !out || frames <= 0 || Ci <= 0 || Hi <= 0 || Wi <= 0 || Co <= 0 || (Co & 15) || Ho <= 0 || Wo <= 0 || qih <= 0 || qiw <= 0 || qch <= 0 ||
        qoh <= 0 || qow <= 0 || sh <= 0 || sw <= 0)
        return -1;
    if ((long)qih * qiw * qch > (long)Ci * Hi * Wi || (long)qoh * qow * qch > (long)Co * Ho * Wo || (long)Co * Ho * Wo > 0x7fffffffL || (long)Ci * Hi * Wi > 0x7fffffffL || (((uintptr_t)out | out_stride) & 15))
        return -1;
    if (Co == Ci && Ho == 2 * Hi && Wo == 2 * Wi && qih == Ci && qiw == Hi && qch == Wi && qoh == Ci && qow == Ho && sh == 2 && sw == 2 && (Ci & 31) == 0 &&
        (((uintptr_t)in | in_stride) & 15) == 0 && !getenv("MARS_HIP_UPSAMPLE_GENERIC")) {
        hipLaunchKernelGGL(upsample_nchwq_tile2_kernel, mv_grid((size_t)Ho * Wo * (Co / 16), frames), dim3(MV_THREADS), 0, mhip_stream_native(), in, in_stride,
                           out, out_stride, Ci, Hi, Wi);
        return mhip_check(hipGetLastError(), "upsample (NCHW-tagged, pixels x channels, 2 x 2)");
    }
    hipLaunchKernelGGL(upsample_nchwq_kernel, mv_grid((size_t)Ho * Wo * (Co / 16), frames), dim3(MV_THREADS), 0, mhip_stream_native(), in, in_stride, Ci,
                       Hi * Wi, out, out_stride, Co, Ho, Wo, qih, qiw, qch, qoh, qow, sh, sw);
    return mhip_check(hipGetLastError(), "upsample (NCHW-tagged, pixels x channels)");
}

// max-pool, stride 1, output size = input size (SPPF's pools): the window runs over CHANNELS c .. c + kh - 1 and map rows h .. h + kw - 1 of
// one column w (clipped at C and H; no padding, identity -128).  A thread owns 16 channels of one pixel: per window row it loads the
// channels c0 .. c0 + 15 + kh - 1 (two 16-byte loads), takes the maximum over the rows per channel, then the sliding maximum over channels.
// KW > 0: the window's row count at compile time -- all its loads are issued before the first maximum (a row past the map is the last row again:
// a maximum does not mind a repeated operand).  With the row loop left to run time every iteration waited for its own two loads: 39 us per
// launch on the 20 x 20 x 128 SPPF maps of yolov5n_int8.mars at batch 256, all of it memory latency.
template <int KW>
__global__ __launch_bounds__(MV_THREADS) void maxpool_nchwq_kernel(const int8_t *in, size_t is, int8_t *out, size_t os, int C, int H, int W, int kh, int kw) {
    const int cg = C / 16;
    const size_t idx = (size_t)blockIdx.x * MV_THREADS + threadIdx.x;
    if (idx >= (size_t)H * W * cg) return;
    const int c0 = (int)(idx % cg) * 16, pix = (int)(idx / cg), h = pix / W;
    const int8_t *s = in + (size_t)blockIdx.y * is + (size_t)pix * C + c0;
    int col[32]; // maximum over the window's rows, channels c0 .. c0 + 31 (kh <= 17)
#pragma unroll
    for (int e = 0; e < 32; e++) col[e] = -128;
    const bool two = c0 + 16 < C;
    const v4i none = (v4i){(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
    if (KW > 0) {
        v4i a[KW > 0 ? KW : 1], b[KW > 0 ? KW : 1];
#pragma unroll
        for (int dy = 0; dy < KW; dy++) {
            const int8_t *r = s + (size_t)(h + dy < H ? dy : H - 1 - h) * W * C;
            a[dy] = *(const v4i *)r;
            b[dy] = two ? *(const v4i *)(r + 16) : none;
        }
#pragma unroll
        for (int dy = 0; dy < KW; dy++) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int va = (int)(int8_t)(a[dy][e >> 2] >> (8 * (e & 3))), vb = (int)(int8_t)(b[dy][e >> 2] >> (8 * (e & 3)));
                col[e] = col[e] > va ? col[e] : va;
                col[16 + e] = col[16 + e] > vb ? col[16 + e] : vb;
            }
        }
    } else {
        for (int dy = 0; dy < kw && h + dy < H; dy++) {
            const int8_t *r = s + (size_t)dy * W * C;
            const v4i a = *(const v4i *)r;
            const v4i b = two ? *(const v4i *)(r + 16) : none;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int va = (int)(int8_t)(a[e >> 2] >> (8 * (e & 3))), vb = (int)(int8_t)(b[e >> 2] >> (8 * (e & 3)));
                col[e] = col[e] > va ? col[e] : va;
                col[16 + e] = col[16 + e] > vb ? col[16 + e] : vb;
            }
        }
    }
    uint32_t wd[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 16; e++) {
        int v = -128;
        if (KW == 5 && kh == 5) { // (the shipped SPPF window: a fixed five-term maximum; columns past C hold -128 already)
#pragma unroll
            for (int dc = 0; dc < 5; dc++) v = v > col[e + dc] ? v : col[e + dc];
        } else {
            for (int dc = 0; dc < kh && c0 + e + dc < C; dc++) v = v > col[(e + dc) & 31] ? v : col[(e + dc) & 31]; // (e + dc <= 31: kh <= 17)
        }
        wd[e >> 2] |= (uint32_t)(uint8_t)v << (8 * (e & 3));
    }
    *(v4i *)(out + (size_t)blockIdx.y * os + (size_t)pix * C + c0) = (v4i){(int)wd[0], (int)wd[1], (int)wd[2], (int)wd[3]};
}

extern "C" int mhip_maxpool_nchwq(const int8_t *in, size_t in_stride, int8_t *out, size_t out_stride, int frames, int C, int H, int W, int kh, int kw) {
    if (!in || !out || frames <= 0 || C <= 0 || (C & 15) || H <= 0 || W <= 0 || kh < 1 || kh > 17 || kw < 1) return -1;
    if ((long)C * H * W > 0x7fffffffL || (((uintptr_t)in | (uintptr_t)out | in_stride | out_stride) & 15)) return -1;
    static const int generic = getenv("MARS_HIP_MAXPOOL_Q_GENERIC") != nullptr; // (A / B switch)
    if (kw == 5 && !generic)
        hipLaunchKernelGGL(maxpool_nchwq_kernel<5>, mv_grid((size_t)H * W * (C / 16), frames), dim3(MV_THREADS), 0, mhip_stream_native(), in, in_stride, out,
                           out_stride, C, H, W, kh, kw);
    else
        hipLaunchKernelGGL(maxpool_nchwq_kernel<0>, mv_grid((size_t)H * W * (C / 16), frames), dim3(MV_THREADS), 0, mhip_stream_native(), in, in_stride, out,
                           out_stride, C, H, W, kh, kw);
    return mhip_check(hipGetLastError(), "maxpool (NCHW-tagged, pixels x channels)");
}

// ---------------------------------------------------------------- upsample
template <int VEC>
__global__ __launch_bounds__(MV_THREADS) void upsample_kernel(const int8_t *in, size_t is, int8_t *out, size_t os,
                                                              int in_h, int in_w, int ch, int out_h, int out_w,
                                                              int scale_h, int scale_w, int pstride, int choff) {
    const int cv = ch / VEC;
    size_t idx = (size_t)blockIdx.x * MV_THREADS + threadIdx.x;
    if (idx >= (size_t)out_h * out_w * cv) return;
    int c = (int)(idx % cv) * VEC;
    size_t pix = idx / cv;
    int ox = (int)(pix % out_w), oy = (int)(pix / out_w);
    int iy = oy / scale_h, ix = ox / scale_w;
    if (iy >= in_h) iy = in_h - 1;
    if (ix >= in_w) ix = in_w - 1;
    const int8_t *s = in + (size_t)blockIdx.y * is + ((size_t)iy * in_w + ix) * ch + c;
    int8_t *d = out + (size_t)blockIdx.y * os + pix * pstride + choff + c;
    if (VEC == 16) *(v4i *)d = *(const v4i *)s;
    else if (VEC == 4) *(int *)d = *(const int *)s;
    else d[0] = s[0];
}

extern "C" int mhip_upsample_i8(const int8_t *in, size_t in_stride, int8_t *out, size_t out_stride, int frames,
                                int in_h, int in_w, int ch, int out_h, int out_w, int scale_h, int scale_w,
                                int out_pix_stride, int out_ch_off) {
    if (!in || !out || frames <= 0 || in_h <= 0 || in_w <= 0 || ch < 0 || out_h < 0 || out_w < 0 || scale_h <= 0 ||
        scale_w <= 0 || out_pix_stride < 0 || out_ch_off < 0)
        return -1;
    if (out_pix_stride && out_pix_stride < out_ch_off + ch) return -1;
    const int pstride = out_pix_stride ? out_pix_stride : ch, choff = out_ch_off;
    size_t total = (size_t)out_h * out_w * ch;
    if (total == 0) return 0;
    bool v16 = (ch % 16 == 0) && ((((uintptr_t)in | (uintptr_t)out | in_stride | out_stride | pstride | choff) & 15) == 0);
    const bool v4 = (ch % 4 == 0) && ((((uintptr_t)in | (uintptr_t)out | in_stride | out_stride | pstride | choff) & 3) == 0);
    if (v16)
        hipLaunchKernelGGL((upsample_kernel<16>), mv_grid(total / 16, frames), dim3(MV_THREADS), 0,
                           mhip_stream_native(), in, in_stride, out, out_stride, in_h, in_w, ch, out_h, out_w,
                           scale_h, scale_w, pstride, choff);
    else if (v4)
        hipLaunchKernelGGL((upsample_kernel<4>), mv_grid(total / 4, frames), dim3(MV_THREADS), 0,
                           mhip_stream_native(), in, in_stride, out, out_stride, in_h, in_w, ch, out_h, out_w,
                           scale_h, scale_w, pstride, choff);
    else
        hipLaunchKernelGGL((upsample_kernel<1>), mv_grid(total, frames), dim3(MV_THREADS), 0, mhip_stream_native(),
                           in, in_stride, out, out_stride, in_h, in_w, ch, out_h, out_w, scale_h, scale_w, pstride, choff);
    return mhip_check(hipGetLastError(), "upsample");
}

// ------------------------------------------------- [C][HW] -> [HW][c_pad]
__global__ __launch_bounds__(MV_THREADS) void nchw_to_nhwc_kernel(const int8_t *in, size_t is, int8_t *out, size_t os,
                                                                  int c, int hw, int c_pad) {
    const int groups = c_pad / 16;
    size_t idx = (size_t)blockIdx.x * MV_THREADS + threadIdx.x;
    if (idx >= (size_t)hw * groups) return;
    // consecutive threads -> consecutive pixels: the strided channel reads coalesce
    int pix = (int)(idx % hw);
    int g = (int)(idx / hw);
    const int8_t *s = in + (size_t)blockIdx.y * is + pix;
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 16; e++) {
        int ci = g * 16 + e;
        uint32_t b = ci < c ? (uint8_t)s[(size_t)ci * hw] : 0u;
        w[e >> 2] |= b << (8 * (e & 3));
    }
    v4i v = {(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
    *(v4i *)(out + (size_t)blockIdx.y * os + (size_t)pix * c_pad + g * 16) = v;
}

// 4 x 4 byte transpose: r[e] = 4 consecutive pixels of channel e  ->  t[p] = channels 0..3 of pixel p (v_perm_b32: 8 per block)
__device__ __forceinline__ void tr4x4(const uint32_t (&r)[4], uint32_t (&t)[4]) {
    const uint32_t lo_ab = __builtin_amdgcn_perm(r[1], r[0], 0x05010400u), hi_ab = __builtin_amdgcn_perm(r[1], r[0], 0x07030602u);
    const uint32_t lo_cd = __builtin_amdgcn_perm(r[3], r[2], 0x05010400u), hi_cd = __builtin_amdgcn_perm(r[3], r[2], 0x07030602u);
    t[0] = __builtin_amdgcn_perm(lo_cd, lo_ab, 0x05040100u);
    t[1] = __builtin_amdgcn_perm(lo_cd, lo_ab, 0x07060302u);
    t[2] = __builtin_amdgcn_perm(hi_cd, hi_ab, 0x05040100u);
    t[3] = __builtin_amdgcn_perm(hi_cd, hi_ab, 0x07060302u);
}

// Round 6: the same relayout at dword granularity (hw % 4 == 0, 4-byte aligned frames: every map of the shipped files).  The byte-wise
// kernel above read 16 single bytes per 16-byte store and wrote 16-byte pieces c_pad apart: 1.7 TB/s of its read + write bytes, and
// 44 % of the GPU time of yolov5n_int8.mars at batch 256 (rocprofv3, profiles/r06_shipped_*).  Here a thread owns FOUR consecutive
// pixels: per group of 16 channels it loads 16 dwords (a wave reads 256 contiguous bytes of each channel row), transposes them with
// v_perm_b32 (4 blocks of 4 x 4 bytes) and stores 16 bytes to each of its 4 pixel rows; the groups of a pixel are walked by the SAME
// thread back to back, so the 16-byte pieces of a c_pad-byte pixel row arrive together and merge in L2.
__global__ __launch_bounds__(MV_THREADS) void nchw_to_nhwc4_kernel(const int8_t *in, size_t is, int8_t *out, size_t os, int c, int hw, int c_pad) {
    const size_t q = (size_t)blockIdx.x * MV_THREADS + threadIdx.x; // 4-pixel group
    if (q * 4 >= (size_t)hw) return;
    const int8_t *s = in + (size_t)blockIdx.y * is + q * 4;
    int8_t *d = out + (size_t)blockIdx.y * os + q * 4 * (size_t)c_pad;
    for (int g = 0; g < c_pad; g += 16) {
        uint32_t r[16];
#pragma unroll
        for (int e = 0; e < 16; e++) r[e] = g + e < c ? *(const uint32_t *)(s + (size_t)(g + e) * hw) : 0u;
        uint32_t t[4][4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint32_t rb[4] = {r[4 * b], r[4 * b + 1], r[4 * b + 2], r[4 * b + 3]};
            tr4x4(rb, t[b]);
        }
#pragma unroll
        for (int p = 0; p < 4; p++) *(v4i *)(d + (size_t)p * c_pad + g) = (v4i){(int)t[0][p], (int)t[1][p], (int)t[2][p], (int)t[3][p]};
    }
}

// [C <= 4][HW] -> [HW][4] (the small-channel stem's input: every pixel widened to 4 bytes): one 16-byte store per 4 pixels
__global__ __launch_bounds__(MV_THREADS) void nchw_to_nhwc_c4_kernel(const int8_t *in, size_t is, int8_t *out, size_t os, int c, int hw) {
    const size_t q = (size_t)blockIdx.x * MV_THREADS + threadIdx.x;
    if (q * 4 >= (size_t)hw) return;
    const int8_t *s = in + (size_t)blockIdx.y * is + q * 4;
    uint32_t r[4], t[4];
#pragma unroll
    for (int e = 0; e < 4; e++) r[e] = e < c ? *(const uint32_t *)(s + (size_t)e * hw) : 0u;
    tr4x4(r, t);
    *(v4i *)(out + (size_t)blockIdx.y * os + q * 16) = (v4i){(int)t[0], (int)t[1], (int)t[2], (int)t[3]};
}

// c_pad: a multiple of 16, or 4 (c <= 4: the small-channel layout)
extern "C" int mhip_nchw_to_nhwc_pad(const int8_t *in, size_t in_stride, int8_t *out, size_t out_stride, int frames,
                                     int c, int hw, int c_pad) {
    if (!in || !out || frames <= 0 || c <= 0 || hw <= 0 || c_pad < c || ((c_pad & 15) && c_pad != 4)) return -1;
    if ((((uintptr_t)out | out_stride) & 15) != 0) return -1;
    const bool dwords = (hw & 3) == 0 && (((uintptr_t)in | in_stride) & 3) == 0;
    if (c_pad == 4) {
        if (!dwords) return -1; // (the planner only chooses the 4-byte layout for maps it can serve: mars_plan.c plan_conv)
        hipLaunchKernelGGL(nchw_to_nhwc_c4_kernel, mv_grid((size_t)hw / 4, frames), dim3(MV_THREADS), 0, mhip_stream_native(), in, in_stride, out,
                           out_stride, c, hw);
        return mhip_check(hipGetLastError(), "nchw_to_nhwc_c4");
    }
    if (dwords && !getenv("MARS_HIP_NCHW_BYTEWISE")) {
        hipLaunchKernelGGL(nchw_to_nhwc4_kernel, mv_grid((size_t)hw / 4, frames), dim3(MV_THREADS), 0, mhip_stream_native(), in, in_stride, out,
                           out_stride, c, hw, c_pad);
        return mhip_check(hipGetLastError(), "nchw_to_nhwc4");
    }
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, mv_grid((size_t)hw * (c_pad / 16), frames), dim3(MV_THREADS), 0,
                       mhip_stream_native(), in, in_stride, out, out_stride, c, hw, c_pad);
    return mhip_check(hipGetLastError(), "nchw_to_nhwc");
}
