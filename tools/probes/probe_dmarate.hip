#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// probe: sustained L2 -> LDS DMA rate per CU for different bytes-per-row patterns (data set small enough to stay in L2)
//   mode 64:  4 lanes fetch one 64-byte row piece (rows 128 B apart... like a 64-byte K step of a 128-channel pixel)
//   mode 128: 8 lanes fetch one 128-byte row (a whole 128-channel pixel)
//   mode 256: 16 lanes fetch 256 contiguous bytes
__device__ __forceinline__ void glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}
template <int LPR> // lanes per row
__global__ __launch_bounds__(512) void k(const int8_t *src, size_t row_stride, int rows_total, int iters, int *sink) {
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // each wave instruction: 64 lanes -> 64 / LPR rows of LPR * 16 bytes
    const int rpi = 64 / LPR;
    int acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 6; j++) { // 6 instructions per wave per step = 48 KB per 8-wave workgroup and step
            const int row = ((blockIdx.x * 97 + it * 131 + (wv * 6 + j) * rpi + lane / LPR) * 7) % rows_total;
            const int koff = ((it + j) % (128 / (LPR * 16) > 0 ? 128 / (LPR * 16) : 1)) * LPR * 16; // slide over the row's 128 bytes
            glds16(src + (size_t)row * row_stride + koff + (lane % LPR) * 16, lds + ((wv * 6 + j) * 1024) + (it & 1) * 49152);
        }
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); // one step in flight behind the current one
        acc += lds[(tid * 16 + it) & 8191];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 123456789) sink[0] = acc;
}
template <int LPR>
static void run(const int8_t *d, int rows, size_t stride, int *sink) {
    hipFuncSetAttribute((const void *)k<LPR>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    const int iters = 400, grid = 256; // one workgroup per CU... then 2
    for (int wgs = 1; wgs <= 1; wgs++) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k<LPR>, dim3(grid * wgs), dim3(512), 98304, 0, d, stride, rows, 20, sink);
        hipEventRecord(a);
        hipLaunchKernelGGL(k<LPR>, dim3(grid * wgs), dim3(512), 98304, 0, d, stride, rows, iters, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double bytes = (double)grid * wgs * iters * 8 * 6 * 1024;
        printf("lanes/row %2d (%3d B per request): %.1f us, %.2f TB/s, %.1f B/clk/CU (2.4 GHz)\n", LPR, LPR * 16, ms * 1e3, bytes / ms / 1e9,
               bytes / (ms * 1e-3) / 256 / 2.4e9);
    }
}
int main() {
    const int rows = 32768; const size_t stride = 128; // 4 MB: L2-resident
    int8_t *d; int *sink; hipMalloc(&d, rows * stride + 4096); hipMemset(d, 1, rows * stride + 4096); hipMalloc(&sink, 64);
    run<4>(d, rows, stride, sink); run<8>(d, rows, stride, sink); run<4>(d, rows, stride, sink); run<8>(d, rows, stride, sink);
    return 0;
}
