// conv_f32_vcat.hip -- the FIRST PIXELS of a 1 x 1 float convolution whose input is a byte-wise CONCAT that is never materialised
// (mars_plan.c virtual_concat_f32; reference mars_runtime.c:971-999 for the concat, src/mars/mxu_conv.c:673-710 for the convolution).  Round 6.
//
// The reference's CONCAT moves runs of shape[3] BYTES whatever the dtype.  On N float maps [1, C, H, W] of equal size that is, in floats
// (run = W / 4, s = (N - 1) * run, per frame):
//     cat[n * run + i] = in_n[i]        i < run, n < N - 1            (the first W bytes of every input but the last)
//     cat[s + j]       = in_last[j]     j < L = C_out * H * W / 4     (the last input, shifted by s floats)
//     cat[j]           = 0              j >= s + L                    (never written: zero_tail_f32)
// A 1 x 1 convolution over cat reads plane c, pixel p at cat[c * HW + p].  For every pixel p >= s that is in_last[c * HW + p - s] for the
// first L / HW planes and zero behind: the SAME convolution over the pointer (in_last - s) with its K loop cut to L / HW planes -- what
// conv_f32_split runs (mhip_conv_f32_t.k_limit), no copy of the concat at all.  What that launch gets wrong is the first s pixels of every
// output plane: there plane 0 holds the other inputs' first bytes (the launch read whatever lies in front of in_last) and plane L / HW still
// holds the tail of in_last's data (the launch stopped one plane early).  This kernel recomputes those s pixels (20 - 60 of 400 - 25 600)
// from their true operands, in float32 with fused multiply-adds, and overwrites them: same stream, after the main launch.
// Bound: nothing -- out_c * s * (L / HW + 1) multiply-adds per frame, a few microseconds.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../mhip.h"

extern "C" hipStream_t mhip_stream_native(void);
extern "C" int mhip_check(hipError_t e, const char *what);

struct vcat_args_t {
    const float *first[3]; // the concat's inputs but the last
    size_t first_stride[3];
    int n_first, run, s;   // their count, floats per run, s = n_first * run
    int hw, planes;        // pixels per plane, planes summed (L / HW + 1)
};

__device__ __forceinline__ float vcat_silu(float v) {
    const float e = __builtin_amdgcn_exp2f(v * -1.44269504088896341f);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}
// bf16, round to nearest even (a NaN stays one): as mhip_conv_f32_split_pack and mars_hip_write_tensor cut records
__device__ __forceinline__ unsigned vcat_bf16(float x) {
    const unsigned b = __float_as_uint(x);
    if ((b & 0x7fffffffu) > 0x7f800000u) return (b >> 16) | 0x40u;
    return (b + 0x7fffu + ((b >> 16) & 1u)) >> 16;
}

// One workgroup (4 waves): 256 output channels x VC_PX consecutive head pixels of one frame.  A lane owns 4 channels x VC_PX pixels; the weights
// come TRANSPOSED ([plane][oc_pitch], packed by the planner: mhip_conv_f32_vcat_pack), so a step is ONE 16-byte weight load and two 16-byte operand
// loads for 32 multiply-adds; the four waves split the planes (c = 1 + wave, 5 + wave, ...: enough waves in flight to hide the loads) and add their
// partial sums through LDS.  What the earlier forms cost, SPPF's reader (15 pixels, 257 planes, 512 channels, 256 frames): a thread per (channel,
// pixel) on the OIHW weights 0.25 ms, a lane per weight row 0.15 ms, a lane per channel on transposed weights with dword loads 0.19 ms -- all of it
// vector-memory instruction issue (9 loads per 8 multiply-adds).
#define VC_PX 8
typedef float vc_f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float vc_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void conv_f32_vcat_head(const mhip_conv_f32_t p, const vcat_args_t a, const int chunks, const int oc_pitch,
                                                          const float *__restrict__ wt) {
    __shared__ float red[4][4 * VC_PX][64];
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6;
    const int oc0 = (int)(blockIdx.x / (unsigned)chunks) * 256 + lane * 4;
    const int px0 = (int)(blockIdx.x % (unsigned)chunks) * VC_PX;
    const unsigned f = blockIdx.y;
    const bool live = oc0 < p.out_c;
    const float *w = wt + (live ? oc0 : 0); // + c * oc_pitch: plane c, channels oc0 .. oc0 + 3 (the image is padded to a multiple of 4)
    const float *x = (const float *)((const char *)p.in + (size_t)f * p.in_stride) + (px0 - a.s); // + c * hw + i: plane c >= 1, pixel px0 + i
    float acc[4][VC_PX];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int i = 0; i < VC_PX; i++) acc[q][i] = 0.0f;
    if (wv == 0) { // plane 0 of these pixels is the other inputs' first bytes (px < s <= hw: always); every other plane the shifted last input
        const vc_f4 w0 = *(const vc_f4 *)w;
#pragma unroll
        for (int i = 0; i < VC_PX; i++) {
            const int px = px0 + i < a.s ? px0 + i : a.s - 1;
            const int n = px / a.run, k = px % a.run;
            const float *fp = n == 0 ? a.first[0] : n == 1 ? a.first[1] : a.first[2];
            const size_t fs = n == 0 ? a.first_stride[0] : n == 1 ? a.first_stride[1] : a.first_stride[2];
            const float v = ((const float *)((const char *)fp + (size_t)f * fs))[k];
#pragma unroll
            for (int q = 0; q < 4; q++) acc[q][i] = w0[q] * v;
        }
    }
    // (pixels behind the last head pixel read what follows in the plane -- at most 7 floats, inside the tensor's allocation -- and are dropped)
#pragma unroll 4
    for (int c = 1 + wv; c < a.planes; c += 4) {
        const vc_f4 wc = *(const vc_f4 *)(w + (size_t)c * oc_pitch);
        const float *xc = x + (size_t)c * a.hw;
        const vc_f4u x0 = *(const vc_f4u *)xc, x1 = *(const vc_f4u *)(xc + 4);
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                acc[q][i] = __builtin_fmaf(wc[q], x0[i], acc[q][i]);
                acc[q][4 + i] = __builtin_fmaf(wc[q], x1[i], acc[q][4 + i]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int i = 0; i < VC_PX; i++) red[wv][q * VC_PX + i][lane] = acc[q][i];
    __syncthreads();
    // thread (lane, wv) ends with channel oc0 + wv, its VC_PX pixels
    const int oc = oc0 + wv;
    if (oc >= p.out_c) return;
    const float bias = p.bias ? p.bias[oc] : 0.0f;
#pragma unroll
    for (int i = 0; i < VC_PX; i++) {
        const int px = px0 + i;
        if (px >= a.s) break;
        float v = bias;
#pragma unroll
        for (int k = 0; k < 4; k++) v += red[k][wv * VC_PX + i][lane];
        if (p.silu) v = vcat_silu(v);
        if (p.out_rec) { // the two pieces of this channel in the record of pixel px (conv_f32_split.hip says what a record is)
            const unsigned h = vcat_bf16(v);
            const float hv = __uint_as_float(h << 16);
            const float r = hv - hv == 0.0f ? v - hv : 0.0f; // no residual of a non-finite hi
            unsigned short *rec = (unsigned short *)((char *)p.out + (size_t)f * p.out_stride + ((size_t)(oc >> 3) * a.hw + px) * 32u);
            rec[oc & 7] = (unsigned short)h;
            rec[8 + (oc & 7)] = (unsigned short)vcat_bf16(r);
        } else {
            ((float *)((char *)p.out + (size_t)f * p.out_stride))[(size_t)oc * a.hw + px] = v;
        }
    }
}

// the head launch's weight image: the first `planes` input channels of the OIHW (1 x 1) weights, transposed to [plane][oc_pitch], oc_pitch =
// out_c rounded up to 4 (zero filled); returns its bytes
extern "C" size_t mhip_conv_f32_vcat_pack(int out_c, int in_c, int planes, const float *w, float *out) {
    if (out_c <= 0 || in_c <= 0 || planes <= 0 || planes > in_c) return 0;
    const size_t pitch = ((size_t)out_c + 3) & ~(size_t)3, bytes = (size_t)planes * pitch * 4;
    if (!w || !out) return bytes;
    memset(out, 0, bytes);
    for (int c = 0; c < planes; c++)
        for (int oc = 0; oc < out_c; oc++) out[(size_t)c * pitch + oc] = w[(size_t)oc * in_c + c];
    return bytes;
}

// p: the convolution as mars_run.c fills it for the main launch, but p->in = the concat's LAST input itself (not shifted) and p->k_limit = the
// planes to sum (those of the main launch + 1); w_t: mhip_conv_f32_vcat_pack's image for that many planes; first / first_strides / n_first: the
// other inputs, in order; run_floats: floats per run.
extern "C" int mhip_conv_f32_vcat_head(const mhip_conv_f32_t *p, const float *w_t, const float *const *first, const size_t *first_strides, int n_first, int run_floats) {
    if (!p || !p->in || !p->out || !w_t || !first || !first_strides || n_first < 1 || n_first > 3 || run_floats <= 0) return -1;
    if (p->kh != 1 || p->kw != 1 || p->stride_h != 1 || p->stride_w != 1 || p->pad_top || p->pad_left || p->add || p->in_rec) return -1;
    if (p->in_h != p->out_h || p->in_w != p->out_w || p->frames <= 0 || p->frames > 65535 || p->in_c <= 0 || p->out_c <= 0) return -1;
    if (p->out_rec && (p->out_c & 7)) return -1;
    const long hw = (long)p->in_h * p->in_w, s = (long)n_first * run_floats;
    if (s >= hw || hw > 0x7fffffffL || p->k_limit < 1 || p->k_limit > p->in_c) return -1;
    vcat_args_t a;
    for (int i = 0; i < 3; i++) {
        a.first[i] = i < n_first ? first[i] : nullptr;
        a.first_stride[i] = i < n_first ? first_strides[i] : 0;
        if (i < n_first && !first[i]) return -1;
    }
    a.n_first = n_first; a.run = run_floats; a.s = (int)s; a.hw = (int)hw; a.planes = p->k_limit;
    const int chunks = (int)((s + VC_PX - 1) / VC_PX), oc_pitch = (p->out_c + 3) & ~3;
    hipLaunchKernelGGL(conv_f32_vcat_head, dim3((unsigned)((p->out_c + 255) / 256) * (unsigned)chunks, (unsigned)p->frames), dim3(256), 0, mhip_stream_native(),
                       *p, a, chunks, oc_pitch, w_t);
    return mhip_check(hipGetLastError(), "conv_f32_vcat_head");
}
