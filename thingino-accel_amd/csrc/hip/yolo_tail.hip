// yolo_tail.hip -- detection tail on gfx950: ordered decode of [npred][85]
// int8 predictions and the reference's exchange-sort + greedy class-wise IoU
// suppression, one workgroup per frame, wavefront-wide scans instead of the
// reference's serial loops.
//
// Replaces reference src/mars/mars_yolo_test.c:80-104 (parse_output) and
// :107-130 (nms).  Bit-exactness notes:
//  * every expf() the reference evaluates has an int8-derived argument, so the
//    host tabulates val[q] = (float)q*scale, obj[q] = 1/(1+expf(-(float)q*scale))
//    and den[q] = 1+expf(-val[q]) with ITS libm; the GPU only divides
//    (correctly rounded) and compares;
//  * candidates keep ascending prediction order and stop at 1000;
//  * the exchange sort `for i: for j>i: if d[j].conf > d[i].conf swap` is NOT
//    stable; pass i moves the suffix maximum to slot i and shifts the chain of
//    strict left-to-right records one record-slot down.  Each pass is done here
//    as one wave-wide (max, first-holder) scan over register-resident slots,
//    reproducing the permutation exactly, ties included;
//  * suppression: the pair relation is evaluated in parallel, 64 rows at a time, into an LDS bit
//    matrix with the reference's float expression order, then applied greedily.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../mhip.h"

extern "C" hipStream_t mhip_stream_native(void);
extern "C" int mhip_check(hipError_t e, const char *what);

#define MAXD 1000
#define ROW 85
#define NCLS 80
#define CONF_MIN 0.25f

struct det_rec {
    float x, y, w, h, conf;
    int cls;
};

// ------------------------------------------------------------------ decode
__global__ __launch_bounds__(256) void decode_kernel(const mhip_detect_t p) {
    __shared__ int wave_cnt[4];
    __shared__ int run_total;
    const int f = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    det_rec *dets = (det_rec *)p.dets + (size_t)f * MAXD;
    if (tid == 0) run_total = 0;
    __syncthreads();
    int total = 0;
    for (int sgi = 0; sgi < p.nseg && total < MAXD; sgi++) {
        const int8_t *pred = p.pred[sgi] + (size_t)f * p.stride[sgi];
        const float *val = p.lut[sgi], *obj = val + 256, *den = val + 512;
        const int np = p.npred[sgi];
        const int pc = p.pix_c[sgi], ps = p.pix_stride[sgi], apx = (ps && pc % ROW == 0) ? pc / ROW : 0;
        for (int base = 0; base < np && total < MAXD; base += 256) {
            const int r = base + tid;
            bool cand = false;
            det_rec d;
            if (r < np) {
                // padded pixel rows (the model wrote its 255-channel rows at a 256-byte pitch): prediction r is anchor
                // r % apx of pixel r / apx; a channel count that is no multiple of 85 takes the per-byte mapping
                const int8_t *rowp = pred + (size_t)r * ROW;
                if (ps && apx) rowp = pred + (size_t)(r / apx) * ps + (r % apx) * ROW;
                const bool bytewise = ps && !apx;
                auto at = [&](int k) -> int {
                    if (!bytewise) return rowp[k];
                    const size_t o = (size_t)r * ROW + k;
                    return pred[(o / pc) * ps + (o % pc)];
                };
                float o = obj[at(4) + 128];
                if (!(o < CONF_MIN)) {
                    int arg = 0, argq = -1000;
                    float top = -1e9f;
                    for (int c = 0; c < NCLS; c++) {
                        int q = at(5 + c);
                        float s = val[q + 128];
                        if (s > top) { top = s; arg = c; argq = q; }
                    }
                    // 1 + expf(-top); with no class above -1e9 the reference divides by +inf
                    float dn = argq == -1000 ? INFINITY : den[argq + 128];
                    float conf = o / dn;
                    if (!(conf < CONF_MIN)) {
                        cand = true;
                        d.x = val[at(0) + 128];
                        d.y = val[at(1) + 128];
                        d.w = val[at(2) + 128];
                        d.h = val[at(3) + 128];
                        d.conf = conf;
                        d.cls = arg;
                    }
                }
            }
            unsigned long long m = __ballot(cand);
            int before = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) wave_cnt[wv] = __popcll(m);
            __syncthreads();
            int off = total;
            for (int w = 0; w < wv; w++) off += wave_cnt[w];
            int slot = off + before;
            if (cand && slot < MAXD) dets[slot] = d;
            int step = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
            total += step;
            __syncthreads();
        }
    }
    if (total > MAXD) total = MAXD;
    if (tid == 0) {
        if (p.raw_counts) p.raw_counts[f] = total;
        p.counts[f] = total;
    }
}

// ------------------------------------------------------------------- sort
// The whole candidate list of a frame sits in REGISTERS (thread T owns slots 4T..4T+3).  Pass i of the
// reference's exchange sort = one (max, first holder) scan over slots >= i, done with DPP row shifts / broadcasts
// inside a wave and one LDS hand-off between the waves:
//   every strict left-to-right record receives the previous record's element,
//   slot i receives the suffix maximum (first occurrence).
// After pass i slot i is final, so when the loop ends the registers hold the
// permutation the reference produces, ties included.
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143
#define DPP_WAVE_SHR1 0x138

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void scan_step(float &v, int &id) {
    const int ninf = __float_as_int(-INFINITY);
    const float ov = __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
    const int oid = __builtin_amdgcn_update_dpp(-1, id, CTRL, ROW_MASK, 0xf, false);
    const bool keep = v > ov; // the earlier holder stays unless strictly beaten
    v = keep ? v : ov;
    id = keep ? id : oid;
}

// 4 waves per frame (256 threads, 4 slots per lane: position = 4 * tid + k).  What the tail costs the overlapped
// convolutions follows how LONG it runs, so a pass is spread over 4 SIMDs: a wave scans its 256 positions, the four
// wave totals (maximum, first holder) meet in LDS (ping-pong by pass parity: one barrier per pass), and every lane
// folds the totals of the waves before its own into its running (maximum, holder).  (One wave with 16 slots per
// lane and no barrier: 0.83 instead of 0.74 ms for the whole tail of 256 frames x 1000 candidates.)
// A slot that is final holds confidence -inf from then on (only its record id is needed), so no pass has to test
// "position >= i": finished slots can neither win the maximum nor be records.  Slot i itself turns into -inf by the
// chain shift of pass i (it is the first record, the running maximum before it is -inf), and receives the maximum's
// id explicitly.  Passes are unrolled by 4 so that the head slot index is a compile-time constant.
__global__ __launch_bounds__(256) void sort_kernel(det_rec *all, const int *counts) {
    __shared__ float tot_v[2][4];
    __shared__ int tot_id[2][4];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    det_rec *dets = all + (size_t)f * MAXD;
    int n = counts[f];
    if (n > MAXD) n = MAXD;
    if (n <= 1) return;
    float c[4];
    int d[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int pos = tid * 4 + k;
        c[k] = pos < n ? dets[pos].conf : -INFINITY;
        d[k] = pos;
    }
    int par = 0;
    for (int i0 = 0; i0 + 1 < n; i0 += 4) {
        const bool owner = tid == (i0 >> 2);
#pragma unroll
        for (int k0 = 0; k0 < 4; k0++) {
            if (i0 + k0 + 1 >= n) break; // uniform
            // (1) this lane's (max, first holder)
            float v = c[0];
            int id = d[0];
#pragma unroll
            for (int k = 1; k < 4; k++) {
                const bool take = c[k] > v;
                v = take ? c[k] : v;
                id = take ? d[k] : id;
            }
            // (2) inclusive scan across the wave; lane 63 publishes the wave's total
            scan_step<DPP_ROW_SHR(1), 0xf>(v, id);
            scan_step<DPP_ROW_SHR(2), 0xf>(v, id);
            scan_step<DPP_ROW_SHR(4), 0xf>(v, id);
            scan_step<DPP_ROW_SHR(8), 0xf>(v, id);
            scan_step<DPP_ROW_BCAST15, 0xa>(v, id);
            scan_step<DPP_ROW_BCAST31, 0xc>(v, id);
            if (lane == 63) { tot_v[par][wv] = v; tot_id[par][wv] = id; }
            float rv = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-INFINITY), __float_as_int(v), DPP_WAVE_SHR1, 0xf, 0xf, false));
            int rid = __builtin_amdgcn_update_dpp(-1, id, DPP_WAVE_SHR1, 0xf, 0xf, false);
            __syncthreads();
            // (3) totals of the waves before this one come first (earlier positions): they keep the record unless
            // strictly beaten; all four in order give the global maximum's first holder
            float av = -INFINITY, gv = -INFINITY;
            int aid = -1, gid = -1;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const float tv = tot_v[par][w];
                const int ti = tot_id[par][w];
                if (w < wv) { const bool t = tv > av; av = t ? tv : av; aid = t ? ti : aid; }
                const bool g = tv > gv; gv = g ? tv : gv; gid = g ? ti : gid;
            }
            {
                const bool keep = rv > av;
                rv = keep ? rv : av;
                rid = keep ? rid : aid;
            }
            par ^= 1;
            // (4) every strict left-to-right record takes the previous record's element
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool rec = c[k] > rv;
                const float tv = c[k];
                const int td = d[k];
                c[k] = rec ? rv : tv;
                d[k] = rec ? rid : td;
                rv = rec ? tv : rv;
                rid = rec ? td : rid;
            }
            d[k0] = owner ? gid : d[k0];
        }
    }
    det_rec out[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (tid * 4 + k < n) out[k] = dets[d[k]];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (tid * 4 + k < n) dets[tid * 4 + k] = out[k];
}

// ---- the same permutation as a systolic pipeline (default).  The exchange sort is a fixed sequence of
// compare-exchanges CE(i, j), i < j, "if conf[j] > conf[i] swap" in lexicographic order: a sorting NETWORK, and
// comparators that touch disjoint positions commute.  Read as hardware it is a linear array of n cells: cell i keeps
// position i, takes the first element that reaches it, compare-exchanges (strict >) with every later arrival and
// passes the loser on; what cell i-1 emits, in order, is exactly what the sequential pass i sees.  One wave per frame:
// lane L holds cells L*CPL .. L*CPL+CPL-1 in registers, every step one element enters lane 0, walks the lane's cells
// (CPL dependent compare-exchanges) and is handed to lane L+1 by one wave-wide DPP shift.  n + n/CPL steps of
// ~5*CPL + 8 vector instructions: for 1000 candidates 94k instructions on ONE wave, against 160k on each of the four
// waves of the pass-by-pass form -- 7x less issue work beside the next batch's convolutions, and no barrier.
// Empty cells hold (-inf, -1): the first real arrival beats it and the sentinel it emits beats nothing downstream.
template <int CPL>
__device__ __forceinline__ void systolic_sort(const det_rec *__restrict__ dets, int n, int lane, unsigned short *__restrict__ perm) {
    float hv[CPL];
    int hid[CPL];
#pragma unroll
    for (int k = 0; k < CPL; k++) { hv[k] = -INFINITY; hid[k] = -1; }
    const int steps = n + (n + CPL - 1) / CPL; // the last element enters at step n-1 and reaches the last lane in use
    float ov = -INFINITY;                      // what this lane emitted in the previous step
    int oid = -1;
    float feed = dets[0].conf; // wave-uniform address: a scalar load, one step ahead of its use
    for (int t = 0; t < steps; t++) {
        const float nextfeed = dets[t + 1 < n ? t + 1 : n - 1].conf;
        float iv = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-INFINITY), __float_as_int(ov), DPP_WAVE_SHR1, 0xf, 0xf, false));
        int iid = __builtin_amdgcn_update_dpp(-1, oid, DPP_WAVE_SHR1, 0xf, 0xf, false);
        if (lane == 0) { iv = t < n ? feed : -INFINITY; iid = t; }
#pragma unroll
        for (int k = 0; k < CPL; k++) {
            const bool gt = iv > hv[k];
            const float kv = gt ? iv : hv[k];
            const int ki = gt ? iid : hid[k];
            iv = gt ? hv[k] : iv;
            iid = gt ? hid[k] : iid;
            hv[k] = kv;
            hid[k] = ki;
        }
        ov = iv;
        oid = iid;
        feed = nextfeed;
    }
#pragma unroll
    for (int k = 0; k < CPL; k++)
        if (lane * CPL + k < n) perm[lane * CPL + k] = (unsigned short)hid[k];
}

// No LDS and ~50 registers, so that the wave fits beside any convolution workgroup; the records themselves are not
// moved: perm[f][pos] = original index of the record that the reference's sort leaves at `pos`, applied by the NMS
// kernel when it loads the boxes.
__global__ __launch_bounds__(64) void sort_systolic_kernel(const det_rec *__restrict__ all, const int *__restrict__ counts,
                                                           unsigned short *__restrict__ perm_all) {
    const int f = blockIdx.x, lane = threadIdx.x;
    const det_rec *dets = all + (size_t)f * MAXD;
    unsigned short *perm = perm_all + (size_t)f * 1024;
    int n = counts[f];
    if (n > MAXD) n = MAXD;
    if (n <= 0) return;
    if (n == 1) { if (lane == 0) perm[0] = 0; return; }
    const int need = (n + 63) >> 6; // cells per lane
    if (need <= 1) systolic_sort<1>(dets, n, lane, perm);
    else if (need <= 2) systolic_sort<2>(dets, n, lane, perm);
    else if (need <= 4) systolic_sort<4>(dets, n, lane, perm);
    else if (need <= 8) systolic_sort<8>(dets, n, lane, perm);
    else systolic_sort<16>(dets, n, lane, perm);
}

// --------------------------------------------------------------- suppress
// One 256-thread workgroup per frame, 32 KB of LDS, so that it shares a CU with the convolution workgroups of
// the NEXT batch (the tail runs on the auxiliary stream; a 1024-thread / 148 KB version of this kernel evicted
// every convolution workgroup from the chip while it ran).  The pairwise "same class and IoU > t" relation is
// evaluated 64 rows at a time, inside class buckets, into an 8 KB bit matrix; wave 0 then walks those rows
// greedily OR-ing them into the removed set (suppressed boxes suppress nothing) -- exactly the reference's
// double loop.  The float expression order of the IoU is the reference's.
#define NMS_THREADS 512
#define NMS_SUBS (NMS_THREADS / NMS_CHUNK) // threads that share a row's bucket
#define NMS_CHUNK 64
#define NMS_BUCKETS 128
__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(det_rec *all, int *counts, float thresh, const unsigned short *perm_all) {
    __shared__ float bx[1024], by[1024], bw[1024], bh[1024], bconf[1024];
    __shared__ int bc[1024];
    __shared__ unsigned short blist[1024];              // box indices grouped by class bucket
    __shared__ int bstart[NMS_BUCKETS + 1], bfill[NMS_BUCKETS];
    __shared__ unsigned int mask[NMS_CHUNK][32];        // 64 rows x 1024 bits
    __shared__ unsigned long long removed_s[16];
    __shared__ int wave_cnt[NMS_THREADS / 64];

    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    det_rec *dets = all + (size_t)f * MAXD;
    int n = counts[f];
    if (n > MAXD) n = MAXD;
    if (n <= 0) return;
    if (tid < NMS_BUCKETS) bfill[tid] = 0;
    for (int j = tid; j < n; j += NMS_THREADS) { // in the sorted order: through the permutation when the sort left one
        det_rec d = dets[perm_all ? perm_all[(size_t)f * 1024 + j] : j];
        bx[j] = d.x; by[j] = d.y; bw[j] = d.w; bh[j] = d.h; bc[j] = d.cls; bconf[j] = d.conf;
    }
    __syncthreads();
    // Only boxes of the same class can suppress each other (reference :118), so the pair relation is evaluated
    // inside class buckets (class & 127; the exact class is still compared): ~n^2/160 IoUs instead of n^2/2 compares.
    for (int j = tid; j < n; j += NMS_THREADS) atomicAdd(&bfill[(unsigned)bc[j] & (NMS_BUCKETS - 1)], 1);
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int b = 0; b < NMS_BUCKETS; b++) { bstart[b] = acc; acc += bfill[b]; bfill[b] = 0; }
        bstart[NMS_BUCKETS] = acc;
    }
    __syncthreads();
    for (int j = tid; j < n; j += NMS_THREADS) {
        const int b = (unsigned)bc[j] & (NMS_BUCKETS - 1);
        blist[bstart[b] + atomicAdd(&bfill[b], 1)] = (unsigned short)j;
    }
    __syncthreads();
    const int nw = (n + 63) >> 6;
    unsigned long long removed = 0; // wave 0: lane w (< 16) holds word w of the removed set
    for (int i0 = 0; i0 < n; i0 += NMS_CHUNK) {
        const int rows = n - i0 < NMS_CHUNK ? n - i0 : NMS_CHUNK;
        for (int k = tid; k < NMS_CHUNK * 32; k += NMS_THREADS) ((unsigned int *)mask)[k] = 0;
        __syncthreads();
        {
            const int r = tid / NMS_SUBS, sub = tid % NMS_SUBS, i = i0 + r;
            if (r < rows) {
                const float xi = bx[i], yi = by[i], wi = bw[i], hi = bh[i];
                const int ci = bc[i];
                const float ax1 = xi - wi / 2, ay1 = yi - hi / 2, ax2 = xi + wi / 2, ay2 = yi + hi / 2;
                const float aarea = wi * hi;
                const int b = (unsigned)ci & (NMS_BUCKETS - 1);
                for (int e = bstart[b] + sub; e < bstart[b + 1]; e += NMS_SUBS) {
                    const int j = blist[e];
                    if (j <= i || bc[j] != ci) continue;
                    const float xj = bx[j], yj = by[j], wj = bw[j], hj = bh[j];
                    float x1 = fmaxf(ax1, xj - wj / 2);
                    float y1 = fmaxf(ay1, yj - hj / 2);
                    float x2 = fminf(ax2, xj + wj / 2);
                    float y2 = fminf(ay2, yj + hj / 2);
                    float iw = fmaxf(0.0f, x2 - x1), ih = fmaxf(0.0f, y2 - y1);
                    float inter = iw * ih;
                    float barea = wj * hj;
                    float uni = aarea + barea;
                    uni = uni - inter;
                    uni = uni + 1e-6f;
                    if (inter / uni > thresh) atomicOr(&mask[r][j >> 5], 1u << (j & 31));
                }
            }
        }
        __syncthreads();
        if (tid < 64) {
            for (int r = 0; r < rows; r++) {
                const int i = i0 + r;
                const unsigned lo = __builtin_amdgcn_readlane((unsigned)removed, i >> 6);
                const unsigned hi = __builtin_amdgcn_readlane((unsigned)(removed >> 32), i >> 6);
                const unsigned long long word = ((unsigned long long)hi << 32) | lo;
                if ((word >> (i & 63)) & 1ull) continue; // suppressed boxes suppress nothing
                if (tid < nw) removed |= (unsigned long long)mask[r][2 * tid] | ((unsigned long long)mask[r][2 * tid + 1] << 32);
            }
        }
        __syncthreads(); // the next chunk overwrites the bit matrix
    }
    if (tid < 16) removed_s[tid] = tid < nw ? removed : ~0ull;
    __syncthreads();
    // compact the survivors in order (a record only ever moves to a lower slot, 256 at a time)
    int total = 0;
    for (int base = 0; base < n; base += NMS_THREADS) {
        const int idx = base + tid;
        const bool keep = idx < n && !((removed_s[idx >> 6] >> (idx & 63)) & 1ull);
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wave_cnt[wv] = __popcll(m);
        __syncthreads();
        int off = total;
        for (int w = 0; w < wv; w++) off += wave_cnt[w];
        const int slot = off + __popcll(m & ((1ull << lane) - 1ull));
        if (keep) {
            det_rec k;
            k.x = bx[idx]; k.y = by[idx]; k.w = bw[idx]; k.h = bh[idx]; k.conf = bconf[idx]; k.cls = bc[idx];
            dets[slot] = k;
        }
        for (int w = 0; w < NMS_THREADS / 64; w++) total += wave_cnt[w];
        __syncthreads();
    }
    if (tid == 0) counts[f] = total;
}

// permutation buffer of the systolic sort: [frames][1024] indices, grown on demand (a hipMalloc synchronises; it
// happens at the first call and when a larger batch appears)
static unsigned short *g_perm = nullptr;
static int g_perm_frames = 0;
extern "C" void mhip_tail_release(void) {
    if (g_perm) (void)hipFree(g_perm);
    g_perm = nullptr;
    g_perm_frames = 0;
}
static int launch_sort_nms(det_rec *dets, int *counts, int frames, float thresh) {
    static int form = -1; // MARS_HIP_SORT=passes: the pass-by-pass kernel (4 waves per frame); default: the systolic form
    if (form < 0) {
        const char *e = getenv("MARS_HIP_SORT");
        form = (e && !strcmp(e, "passes")) ? 1 : 0;
    }
    const unsigned short *perm = nullptr;
    if (form) {
        hipLaunchKernelGGL(sort_kernel, dim3(frames), dim3(256), 0, mhip_stream_native(), dets, counts);
    } else {
        if (frames > g_perm_frames) {
            if (hipDeviceSynchronize() != hipSuccess) return -1;
            if (g_perm) (void)hipFree(g_perm);
            g_perm = nullptr;
            g_perm_frames = 0;
            if (mhip_check(hipMalloc((void **)&g_perm, (size_t)frames * 1024 * sizeof(unsigned short)), "hipMalloc sort permutation")) return -1;
            g_perm_frames = frames;
        }
        hipLaunchKernelGGL(sort_systolic_kernel, dim3(frames), dim3(64), 0, mhip_stream_native(), dets, counts, g_perm);
        perm = g_perm;
    }
    int rc = mhip_check(hipGetLastError(), "sort");
    if (rc) return rc;
    hipLaunchKernelGGL(nms_kernel, dim3(frames), dim3(NMS_THREADS), 0, mhip_stream_native(), dets, counts, thresh, perm);
    return mhip_check(hipGetLastError(), "nms");
}

extern "C" int mhip_detect(const mhip_detect_t *p) {
    if (!p || p->nseg <= 0 || p->nseg > 4 || p->frames <= 0 || !p->dets || !p->counts) return -1;
    for (int s = 0; s < p->nseg; s++)
        if (!p->pred[s] || !p->lut[s] || p->npred[s] < 0 || (p->pix_stride[s] && (p->pix_c[s] <= 0 || p->pix_stride[s] < p->pix_c[s]))) return -1;
    hipLaunchKernelGGL(decode_kernel, dim3(p->frames), dim3(256), 0, mhip_stream_native(), *p);
    int rc = mhip_check(hipGetLastError(), "decode");
    if (rc || !p->do_nms) return rc;
    return launch_sort_nms((det_rec *)p->dets, p->counts, p->frames, p->nms_thresh);
}

extern "C" int mhip_nms_only(void *dets_dev, int *count_dev, int n, float thresh) {
    if (!dets_dev || !count_dev || n < 0 || n > MAXD) return -1;
    return launch_sort_nms((det_rec *)dets_dev, count_dev, 1, thresh);
}
