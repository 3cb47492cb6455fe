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
//    as one wave-wide (max, first-position) scan, reproducing the permutation
//    exactly, ties included.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

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
        for (int base = 0; base < np && total < MAXD; base += 256) {
            const int r = base + tid;
            bool cand = false;
            det_rec d;
            if (r < np) {
                const int8_t *row = pred + (size_t)r * ROW;
                float o = obj[row[4] + 128];
                if (!(o < CONF_MIN)) {
                    int arg = 0, argq = -1000;
                    float top = -1e9f;
                    for (int c = 0; c < NCLS; c++) {
                        int q = row[5 + c];
                        float s = val[q + 128];
                        if (s > top) { top = s; arg = c; argq = q; }
                    }
                    // 1 + expf(-top); with no class above -1e9 the reference divides by +inf
                    float dn = argq == -1000 ? INFINITY : den[argq + 128];
                    float conf = o / dn;
                    if (!(conf < CONF_MIN)) {
                        cand = true;
                        d.x = val[row[0] + 128];
                        d.y = val[row[1] + 128];
                        d.w = val[row[2] + 128];
                        d.h = val[row[3] + 128];
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

// ---------------------------------------------------------- sort + suppress
// one wave per frame; everything in LDS
__global__ __launch_bounds__(64) void sort_nms_kernel(det_rec *all, int *counts, float thresh) {
    __shared__ float cf[2][1024];
    __shared__ short id[2][1024];
    __shared__ float bx[1024], by[1024], bw[1024], bh[1024];
    __shared__ int bc[1024];
    __shared__ float sconf[1024];
    __shared__ short sid[1024];
    __shared__ unsigned char dead[1024];

    const int f = blockIdx.x, lane = threadIdx.x;
    det_rec *dets = all + (size_t)f * MAXD;
    int n = counts[f];
    if (n > MAXD) n = MAXD;
    if (n <= 0) return;
    for (int j = lane; j < n; j += 64) {
        cf[0][j] = dets[j].conf;
        id[0][j] = (short)j;
    }
    __syncthreads();

    // ---- exchange sort, pass by pass
    int cur = 0;
    for (int i = 0; i + 1 < n; i++) {
        const int len = n - i;
        const int per = (len + 63) >> 6;
        const int b = i + lane * per;
        const int e = b + per < n ? b + per : n;
        // (1) chunk maximum and the first position that attains it
        float mv = -INFINITY;
        int mp = -1;
        for (int j = b; j < e; j++) {
            float v = cf[cur][j];
            if (v > mv) { mv = v; mp = j; }
        }
        // (2) exclusive scan over lanes: running (max, holder) before this chunk
        float sv = mv;
        int sp = mp;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            float ov = __shfl_up(sv, d);
            int op = __shfl_up(sp, d);
            if (lane >= d && !(sv > ov)) { sv = ov; sp = op; } // keep the earlier holder unless strictly greater
        }
        float rv = __shfl_up(sv, 1);
        int rp = __shfl_up(sp, 1);
        if (lane == 0) { rv = -INFINITY; rp = -1; }
        const int fin = __shfl(sp, 63); // holder of the suffix maximum after the whole pass
        // (3) rewrite the suffix: a strict record receives the previous holder's element
        const int nxt = cur ^ 1;
        for (int j = b; j < e; j++) {
            float v = cf[cur][j];
            int src = j;
            if (v > rv) {
                src = rp; // previous record (or -1 for the chain head at slot i)
                rv = v;
                rp = j;
            }
            if (j == i) src = fin; // slot i ends the pass holding the maximum
            cf[nxt][j] = cf[cur][src < 0 ? j : src];
            id[nxt][j] = id[cur][src < 0 ? j : src];
        }
        __syncthreads();
        if (lane == 0) {
            sconf[i] = cf[nxt][i];
            sid[i] = id[nxt][i];
        }
        cur = nxt;
    }
    if (lane == 0) {
        sconf[n - 1] = cf[cur][n - 1];
        sid[n - 1] = id[cur][n - 1];
    }
    __syncthreads();

    // ---- gather boxes in sorted order
    for (int j = lane; j < n; j += 64) {
        det_rec d = dets[sid[j]];
        bx[j] = d.x; by[j] = d.y; bw[j] = d.w; bh[j] = d.h; bc[j] = d.cls;
        dead[j] = 0;
    }
    __syncthreads();

    // ---- greedy suppression, i sequential, j across the wave
    for (int i = 0; i < n; i++) {
        if (dead[i]) continue; // uniform: every lane reads the same byte
        const float xi = bx[i], yi = by[i], wi = bw[i], hi = bh[i];
        const int ci = bc[i];
        const float ax1 = xi - wi / 2, ay1 = yi - hi / 2, ax2 = xi + wi / 2, ay2 = yi + hi / 2;
        const float aarea = wi * hi;
        for (int j = i + 1 + lane; j < n; j += 64) {
            if (dead[j] || bc[j] != ci) continue;
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
            if (inter / uni > thresh) dead[j] = 1;
        }
        __syncthreads();
    }

    // ---- compact survivors, in order, back to global
    int kept = 0;
    for (int base = 0; base < n; base += 64) {
        int j = base + lane;
        bool keep = j < n && !dead[j];
        unsigned long long m = __ballot(keep);
        int slot = kept + __popcll(m & ((1ull << lane) - 1ull));
        if (keep) {
            det_rec d;
            d.x = bx[j]; d.y = by[j]; d.w = bw[j]; d.h = bh[j]; d.conf = sconf[j]; d.cls = bc[j];
            dets[slot] = d;
        }
        kept += __popcll(m);
    }
    if (lane == 0) counts[f] = kept;
}

extern "C" int mhip_detect(const mhip_detect_t *p) {
    if (!p || p->nseg <= 0 || p->nseg > 4 || p->frames <= 0 || !p->dets || !p->counts) return -1;
    for (int s = 0; s < p->nseg; s++)
        if (!p->pred[s] || !p->lut[s] || p->npred[s] < 0) return -1;
    hipLaunchKernelGGL(decode_kernel, dim3(p->frames), dim3(256), 0, mhip_stream_native(), *p);
    int rc = mhip_check(hipGetLastError(), "decode");
    if (rc || !p->do_nms) return rc;
    hipLaunchKernelGGL(sort_nms_kernel, dim3(p->frames), dim3(64), 0, mhip_stream_native(), (det_rec *)p->dets,
                       p->counts, p->nms_thresh);
    return mhip_check(hipGetLastError(), "sort_nms");
}

extern "C" int mhip_nms_only(void *dets_dev, int *count_dev, int n, float thresh) {
    if (!dets_dev || !count_dev || n < 0 || n > MAXD) return -1;
    hipLaunchKernelGGL(sort_nms_kernel, dim3(1), dim3(64), 0, mhip_stream_native(), (det_rec *)dets_dev, count_dev,
                       thresh);
    return mhip_check(hipGetLastError(), "sort_nms");
}
