#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// probe: sustained L2 -> LDS rate per CU, LDS-DMA (buffer/global_load ... lds, 16 B per lane) AGAINST the register path
// (global_load_dwordx4 into VGPRs, ds_write_b128), same addresses, same 48 KB per 8-wave workgroup and step, one step in flight.
// Round 4: conv_f32_split's gather (register path) was seen issuing 48 KB in ~1050 cycles = 46 B/clk/CU while every LDS-DMA
// structure of rounds 2-3 stayed at 21-27 B/clk/CU -- is the register path the faster way into LDS on this part?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_vgprpath tools/probes/probe_vgprpath.hip && /tmp/probe_vgprpath
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}
template <int LPR, bool DMA> // lanes per row
__global__ __launch_bounds__(512) void k(const int8_t *src, size_t row_stride, int rows_total, int iters, int *sink) {
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int rpi = 64 / LPR;
    int acc = 0;
    v4i r[2][6];
    auto addr = [&](int it, int j) {
        const int row = ((blockIdx.x * 97 + it * 131 + (wv * 6 + j) * rpi + lane / LPR) * 7) % rows_total;
        const int koff = ((it + j) % (128 / (LPR * 16) > 0 ? 128 / (LPR * 16) : 1)) * LPR * 16;
        return src + (size_t)row * row_stride + koff + (lane % LPR) * 16;
    };
    if (!DMA) {
#pragma unroll
        for (int j = 0; j < 6; j++) r[0][j] = *(const v4i *)addr(0, j);
    }
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (DMA) {
#pragma unroll
                for (int j = 0; j < 6; j++) glds16(addr(it + h, j), lds + ((wv * 6 + j) * 1024) + (h & 1) * 49152);
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
#pragma unroll
                for (int j = 0; j < 6; j++) r[h ^ 1][j] = *(const v4i *)addr(it + h + 1, j); // next step's loads
#pragma unroll
                for (int j = 0; j < 6; j++) *(v4i *)(lds + ((wv * 6 + j) * 1024) + (h & 1) * 49152 + lane * 16) = r[h][j]; // this step's data into LDS
            }
            acc += lds[(tid * 16 + it + h) & 8191];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 123456789) sink[0] = acc;
}
template <int LPR, bool DMA>
static void run(const int8_t *d, int rows, size_t stride, int *sink, int wgs) {
    hipFuncSetAttribute((const void *)k<LPR, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    const int iters = 400, grid = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<LPR, DMA>), dim3(grid * wgs), dim3(512), 98304 / wgs, 0, d, stride, rows, 20, sink);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<LPR, DMA>), dim3(grid * wgs), dim3(512), 98304 / wgs, 0, d, stride, rows, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)grid * wgs * iters * 8 * 6 * 1024;
    printf("%-13s %d workgroup(s)/CU  lanes/row %2d (%3d B per request): %8.1f us, %5.2f TB/s, %5.1f B/clk/CU (2.4 GHz)\n", DMA ? "LDS-DMA" : "register path", wgs, LPR, LPR * 16,
           ms * 1e3, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
}
int main() {
    const int rows = 32768; const size_t stride = 128; // 4 MB: L2-resident
    int8_t *d; int *sink; hipMalloc(&d, rows * stride + 4096); hipMemset(d, 1, rows * stride + 4096); hipMalloc(&sink, 64);
    for (int rep = 0; rep < 2; rep++) {
        run<4, true>(d, rows, stride, sink, 1); run<4, false>(d, rows, stride, sink, 1);
        run<8, true>(d, rows, stride, sink, 1); run<8, false>(d, rows, stride, sink, 1);
        run<64, true>(d, rows, stride, sink, 1); run<64, false>(d, rows, stride, sink, 1);
    }
    return 0;
}
