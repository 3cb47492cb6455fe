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

typedef int v4i __attribute__((ext_vector_type(4)));

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
                    if (p.mono[sgi] && !bytewise) {
                        // value[q] strictly increasing (the host checked the table): the first maximum of value[q] is the
                        // first maximum of q.  The 80 class bytes come as five 16-byte loads (any alignment is served) and
                        // are compared as integers; one table read at the end restores `top`.
                        v4i w[5];
#pragma unroll
                        for (int k = 0; k < 5; k++) __builtin_memcpy(&w[k], rowp + 5 + 16 * k, 16);
                        int best = -129;
#pragma unroll
                        for (int k = 0; k < 20; k++) {
                            const int d = w[k >> 2][k & 3];
#pragma unroll
                            for (int b = 0; b < 4; b++) {
                                const int q = (d << (24 - 8 * b)) >> 24;
                                const bool gt = q > best;
                                best = gt ? q : best;
                                arg = gt ? 4 * k + b : arg;
                            }
                        }
                        const float s = val[best + 128];
                        if (s > top) { top = s; argq = best; } else arg = 0; // nothing above -1e9: as if no class had been seen
                    } else {
                        for (int c = 0; c < NCLS; c++) {
                            int q = at(5 + c);
                            float s = val[q + 128];
                            if (s > top) { top = s; arg = c; argq = q; }
                        }
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
// The reference orders the candidates with an exchange sort, `for i: for j > i: if d[j].conf > d[i].conf swap`
// (mars_yolo_test.c:108-110): not stable, and the permutation it leaves among equal confidences is part of the
// contract.  Rounds 1-2 replayed its n passes (wave-wide scans, then a systolic array: 0.5-0.7 ms per batch beside the
// convolutions, the longest kernel of the tail).  Round 3 computes the same permutation in closed form:
//   * the result is non-increasing in confidence, so an element's slot is (number of strictly greater elements) + its
//     place inside its tie group;
//   * for a tie group G of value v, let H = the elements > v.  While any of H is left, each pass takes away exactly the
//     first slot of H in the remaining array -- that element is a strict left-to-right record, so it moves on or is the
//     maximum and leaves -- and if members of G stand before it, the first of them (a record too) jumps into that slot,
//     behind the members it passes.  Whatever the values inside H are, G therefore behaves as a QUEUE walked along the
//     original array: a member of G is appended, an element of H moves the queue's front to its back.  Once H is gone
//     the passes pick the group's members left to right without moving the others, so the queue IS the group's order.
// So: one bitonic sort by (confidence descending, original index ascending) -- 55 compare-exchange rounds for 1024 --
// and, only for tie groups, a replay of that queue by the group's first thread: members arrive in index order, between
// two arrivals the queue turns by (the H elements passed) mod (its length); a circular linked list in LDS makes a turn
// one pointer step and an arrival O(1).  Steps per group <= min(|H|, sum of lengths).  Checked against the oracle's
// literal double loop on tie-heavy inputs (tests/test_gpu_kernels.py, tools/fuzz_tail.py).
// A NaN confidence (a NaN scale gives one, and `conf < 0.25` lets it through) is not ordered by `>`: a frame that
// holds one is sorted by the literal double loop on one thread -- slow, exact, and never taken by a real model.

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
__global__ __launch_bounds__(NMS_THREADS) void sort_nms_kernel(det_rec *all, int *counts, float thresh) {
    __shared__ float bx[1024], by[1024], bw[1024], bh[1024], bconf[1024];
    __shared__ int bc[1024];
    __shared__ unsigned short blist[1024];              // box indices grouped by class bucket
    __shared__ int bstart[NMS_BUCKETS + 1], bfill[NMS_BUCKETS];
    __shared__ __attribute__((aligned(16))) unsigned int mask[NMS_CHUNK][32]; // 64 rows x 1024 bits
    __shared__ unsigned long long removed_s[16];
    __shared__ int wave_cnt[NMS_THREADS / 64];
    __shared__ unsigned short sperm[1024];              // sorted slot -> original index
    __shared__ int flags[2];                            // [0] a NaN confidence in this frame, [1] a tie in this frame

    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    det_rec *dets = all + (size_t)f * MAXD;
    int n = counts[f];
    if (n > MAXD) n = MAXD;
    if (n <= 0) return;
    // ---- sort (see "sort" above).  Scratch that the suppression only needs later: keys = the bit matrix (8 KB),
    // queue links = blist, passed-H counts = the class array.
    {
        unsigned long long *keys = (unsigned long long *)&mask[0][0];
        unsigned short *nxt = blist, *gtb = (unsigned short *)bc;
        int P = 2;
        while (P < n) P <<= 1;
        if (tid < 2) flags[tid] = 0;
        __syncthreads();
        bool nan = false;
        for (int r = tid; r < P; r += NMS_THREADS) {
            unsigned long long k = 0; // padding: below every real key (confidences are >= 0.25)
            if (r < n) {
                const float c = dets[r].conf;
                nan |= c != c;
                k = ((unsigned long long)__float_as_uint(c) << 10) | (unsigned)(1023 - r);
            }
            keys[r] = k;
        }
        if (nan) flags[0] = 1;
        __syncthreads();
        if (flags[0]) {
            // the literal double loop on (confidence, index) pairs; float compares, so NaN behaves as in the reference
            if (tid == 0) {
                for (int i = 0; i + 1 < n; i++) {
                    unsigned long long ki = keys[i];
                    for (int j = i + 1; j < n; j++) {
                        const unsigned long long kj = keys[j];
                        if (__uint_as_float((unsigned)(kj >> 10)) > __uint_as_float((unsigned)(ki >> 10))) { keys[i] = kj; keys[j] = ki; ki = kj; }
                    }
                }
            }
            __syncthreads();
            for (int r = tid; r < n; r += NMS_THREADS) sperm[r] = (unsigned short)(1023 - (int)(keys[r] & 1023));
        } else {
            for (int k = 2; k <= P; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int t = tid; t < (P >> 1); t += NMS_THREADS) {
                        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                        const unsigned long long a = keys[lo], b = keys[hi];
                        const bool desc = (lo & k) == 0;
                        if (desc ? a < b : a > b) { keys[lo] = b; keys[hi] = a; }
                    }
                    __syncthreads();
                }
            // slots of untied elements are final; tie groups are replayed below
            bool member[2] = {false, false};
            for (int q = 0, r = tid; r < n; r += NMS_THREADS, q++) {
                const unsigned cb = (unsigned)(keys[r] >> 10);
                const bool tl = r > 0 && (unsigned)(keys[r - 1] >> 10) == cb, tr = r + 1 < n && (unsigned)(keys[r + 1] >> 10) == cb;
                member[q] = tl || tr;
                sperm[r] = (unsigned short)(1023 - (int)(keys[r] & 1023));
            }
            if (member[0] || member[1]) flags[1] = 1;
            __syncthreads();
            if (flags[1]) { // uniform
                // every member: how many greater elements stand before it in the original order
                for (int q = 0, r = tid; r < n; r += NMS_THREADS, q++) {
                    if (!member[q]) continue;
                    const unsigned cb = (unsigned)(keys[r] >> 10);
                    int g0 = r;
                    while (g0 > 0 && (unsigned)(keys[g0 - 1] >> 10) == cb) g0--;
                    const int pos = sperm[r];
                    int cnt = 0;
                    for (int s2 = 0; s2 < g0; s2++) cnt += sperm[s2] < pos;
                    gtb[r] = (unsigned short)cnt;
                }
                __syncthreads();
                unsigned short first[2] = {0, 0};
                int glen[2] = {0, 0};
                for (int q = 0, r = tid; r < n; r += NMS_THREADS, q++) {
                    if (!member[q]) continue;
                    const unsigned cb = (unsigned)(keys[r] >> 10);
                    if (r > 0 && (unsigned)(keys[r - 1] >> 10) == cb) continue; // not the group's first thread
                    // members r .. r+m-1 arrive in original-index order (the sort's second key)
                    int back = r, len = 1;
                    nxt[r] = (unsigned short)r;
                    int prev = gtb[r];
                    int x = r + 1;
                    for (; x < n && (unsigned)(keys[x] >> 10) == cb; x++) {
                        const int g = gtb[x];
                        for (int turn = (g - prev) % len; turn > 0; turn--) back = nxt[back];
                        prev = g;
                        nxt[x] = nxt[back];
                        nxt[back] = (unsigned short)x;
                        back = x;
                        len++;
                    }
                    for (int turn = (r - prev) % len; turn > 0; turn--) back = nxt[back]; // r = the count of greater elements
                    first[q] = nxt[back];
                    glen[q] = len;
                }
                __syncthreads(); // sperm is read (as the members' original indices) before it is rewritten
                for (int q = 0, r = tid; r < n; r += NMS_THREADS, q++) {
                    if (!glen[q]) continue;
                    // walk the queue front to back; the original indices are still in keys
                    int e = first[q];
                    for (int k2 = 0; k2 < glen[q]; k2++) {
                        sperm[r + k2] = (unsigned short)(1023 - (int)(keys[e] & 1023));
                        e = nxt[e];
                    }
                }
            }
        }
        __syncthreads();
    }
    if (tid < NMS_BUCKETS) bfill[tid] = 0;
    for (int j = tid; j < n; j += NMS_THREADS) { // in the sorted order
        det_rec d = dets[sperm[j]];
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
            // the greedy walk is a dependent chain of `rows` steps; its matrix rows do not depend on it, so they are fetched
            // 16 at a time ahead of the steps that use them (one LDS latency per 16 rows instead of one per row) and a
            // suppressed row is skipped by a select, not a branch
            const int wl = tid < 16 ? tid : 15, c = i0 >> 6; // lane w holds word w of the removed set; rows i0.. live in word c
            for (int r0 = 0; r0 < rows; r0 += 16) {
                unsigned long long mrow[16];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int r = r0 + k < NMS_CHUNK ? r0 + k : NMS_CHUNK - 1;
                    mrow[k] = *(const unsigned long long *)&mask[r][2 * wl];
                }
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int r = r0 + k; // bit r of word c: rows >= `rows` hold an all-zero matrix row
                    const unsigned half = r < 32 ? __builtin_amdgcn_readlane((unsigned)removed, c) : __builtin_amdgcn_readlane((unsigned)(removed >> 32), c);
                    const bool dead = (half >> (r & 31)) & 1u; // suppressed boxes suppress nothing
                    removed |= (dead || tid >= nw || r >= rows) ? 0ull : mrow[k];
                }
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

extern "C" void mhip_tail_release(void) {} // (the systolic sort's permutation buffer lived here until round 3)
static int launch_sort_nms(det_rec *dets, int *counts, int frames, float thresh) {
    hipLaunchKernelGGL(sort_nms_kernel, dim3(frames), dim3(NMS_THREADS), 0, mhip_stream_native(), dets, counts, thresh);
    return mhip_check(hipGetLastError(), "sort + nms");
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
