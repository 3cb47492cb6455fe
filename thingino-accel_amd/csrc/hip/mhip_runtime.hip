// mhip_runtime.hip -- device bring-up, memory and stream plumbing behind the
// C-ABI of mhip.h.  Stands where the reference has ioctl/mmap access to
// /dev/soc-nna and the NNDMA descriptor engine (reference src/device.c:133-302,
// src/memory.c:76-196, src/nna_dma.c:130-252): HBM allocations, pinned host
// buffers and async copies on ONE library stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../mhip.h"

static hipStream_t g_stream = nullptr;
static hipStream_t g_aux = nullptr;   // detection tail runs here, beside the next batch's graph
static hipStream_t g_up = nullptr;    // pipelined I/O: host -> HBM copies of the NEXT batch (created on first use)
static hipStream_t g_down = nullptr;  // pipelined I/O: HBM -> host copies of the PREVIOUS batch
static hipStream_t g_seconds[3] = {nullptr, nullptr, nullptr};
 // second compute stream: the other half of a large batch runs here (mars_model.c)
static int g_use_aux = 0;             // current stream of the launchers: 0 main, 1 aux, 2 upload, 3 download, 4 second compute
static int g_ready = 0;
static int g_device = -1;
static char g_err[256] = "";
static void *g_zero_page = nullptr; // 256 zero bytes in HBM: DMA source for out-of-image conv taps

static hipStream_t stream_of(int which) {
    if (which == 1) return g_aux;
    if (which == 2 || which == 3) {
        hipStream_t &st = which == 2 ? g_up : g_down;
        if (!st && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = nullptr;
        return st ? st : g_stream;
    }
    if (which >= 4 && which <= 6) { // same priority as the main stream: the parts of a batch are peers
        hipStream_t &st = g_seconds[which - 4];
        if (!st) {
            int lo = 0, hi = 0;
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = hi = 0;
            if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi) != hipSuccess) st = nullptr;
        }
        return st ? st : g_stream;
    }
    return g_stream;
}
extern "C" hipStream_t mhip_stream_native(void) { return stream_of(g_use_aux); }

extern "C" int mhip_check(hipError_t e, const char *what) {
    if (e == hipSuccess) return 0;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    if (getenv("MARS_VERBOSE")) fprintf(stderr, "mars-hip: %s\n", g_err);
    return -(int)e;
}

extern "C" const char *mhip_last_error(void) { return g_err; }

extern "C" int mhip_init(int device_hint) {
    if (g_ready) return 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        snprintf(g_err, sizeof(g_err), "no HIP device visible");
        return -1;
    }
    if (device_hint >= n) { // a wrong rank -> device mapping must not silently land on another rank's GPU
        snprintf(g_err, sizeof(g_err), "device %d requested, %d visible", device_hint, n);
        return -1;
    }
    int dev = device_hint >= 0 ? device_hint : 0;
    hipDeviceProp_t prop;
    if (mhip_check(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties")) return -1;
    // the kernels are built for gfx950 only; refuse anything else loudly
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_err, sizeof(g_err), "device %d is %s, this build targets gfx950 (MI355X)", dev, prop.gcnArchName);
        return -2;
    }
    if (mhip_check(hipSetDevice(dev), "hipSetDevice")) return -1;
    // Main stream (the graph) and auxiliary stream (the detection tail of the previous batch) must sit on
    // DIFFERENT hardware queues or the tail serialises with the next batch.  HIP multiplexes streams onto a few
    // hardware queues round-robin (GPU_MAX_HW_QUEUES, default 4), so in a process that owns other streams (torch,
    // RCCL) two plain streams can land on the same queue (measured: +0.9 ms per batch).  Streams of different
    // priority come from different queue pools, which guarantees the separation.
    int prio_lo = 0, prio_hi = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) prio_lo = prio_hi = 0;
    if (mhip_check(hipStreamCreateWithPriority(&g_stream, hipStreamNonBlocking, prio_hi), "hipStreamCreate")) return -3;
    if (mhip_check(hipStreamCreateWithPriority(&g_aux, hipStreamNonBlocking, prio_lo), "hipStreamCreate aux")) return -3;
    if (mhip_check(hipMalloc(&g_zero_page, 256), "hipMalloc zero page") ||
        mhip_check(hipMemset(g_zero_page, 0, 256), "hipMemset zero page"))
        return -3;
    g_device = dev;
    g_ready = 1;
    return 0;
}

extern "C" void mhip_tail_release(void); // yolo_tail.hip: the sort's permutation buffer
extern "C" void mhip_shutdown(void) {
    if (!g_ready) return;
    (void)hipStreamSynchronize(g_stream);
    (void)hipStreamSynchronize(g_aux);
    mhip_tail_release();
    (void)hipStreamDestroy(g_stream);
    (void)hipStreamDestroy(g_aux);
    if (g_up) { (void)hipStreamSynchronize(g_up); (void)hipStreamDestroy(g_up); }
    if (g_down) { (void)hipStreamSynchronize(g_down); (void)hipStreamDestroy(g_down); }
    for (int k = 0; k < 3; k++) {
        if (g_seconds[k]) { (void)hipStreamSynchronize(g_seconds[k]); (void)hipStreamDestroy(g_seconds[k]); }
        g_seconds[k] = nullptr;
    }
    g_up = g_down = nullptr;
    g_aux = nullptr;
    g_use_aux = 0;
    if (g_zero_page) (void)hipFree(g_zero_page);
    g_zero_page = nullptr;
    g_stream = nullptr;
    g_ready = 0;
}

extern "C" int mhip_ready(void) { return g_ready; }
extern "C" const void *mhip_zero_page(void) { return g_zero_page; }

extern "C" int mhip_device_info(int *cu_count, int *lds_bytes, int *gfx_version, size_t *hbm_bytes) {
    if (!g_ready) return -1;
    hipDeviceProp_t prop;
    if (mhip_check(hipGetDeviceProperties(&prop, g_device), "hipGetDeviceProperties")) return -1;
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (gfx_version) *gfx_version = atoi(prop.gcnArchName + 3);
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return 0;
}

extern "C" void *mhip_stream(void) { return (void *)g_stream; }

extern "C" int mhip_sync(void) {
    int rc = mhip_check(hipStreamSynchronize(g_stream), "hipStreamSynchronize");
    int rc2 = mhip_check(hipStreamSynchronize(g_aux), "hipStreamSynchronize aux");
    if (g_up && mhip_check(hipStreamSynchronize(g_up), "hipStreamSynchronize upload") && !rc) rc = -1;
    if (g_down && mhip_check(hipStreamSynchronize(g_down), "hipStreamSynchronize download") && !rc) rc = -1;
    for (int k = 0; k < 3; k++)
        if (g_seconds[k] && mhip_check(hipStreamSynchronize(g_seconds[k]), "hipStreamSynchronize second") && !rc) rc = -1;
    return rc ? rc : rc2;
}
// every launcher enqueues on "the current stream": main by default, aux while selected
extern "C" void mhip_select_aux(int on) { g_use_aux = on ? 1 : 0; }
extern "C" void mhip_select_stream(int which) { g_use_aux = which >= 0 && which <= 6 ? which : 0; }
// make a stream (0 main, 1 aux, 2 upload, 3 download) wait for an event recorded elsewhere
extern "C" int mhip_stream_wait(int which, void *ev) {
    return mhip_check(hipStreamWaitEvent(stream_of(which), (hipEvent_t)ev, 0), "hipStreamWaitEvent");
}
extern "C" int mhip_event_sync(void *ev) { return mhip_check(hipEventSynchronize((hipEvent_t)ev), "hipEventSynchronize"); }

extern "C" void *mhip_malloc(size_t bytes) {
    void *p = nullptr;
    if (!g_ready) return nullptr;
    if (mhip_check(hipMalloc(&p, bytes ? bytes : 256), "hipMalloc")) return nullptr;
    return p;
}

extern "C" void mhip_free(void *p) {
    if (p) (void)hipFree(p);
}

extern "C" void *mhip_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (!g_ready) return nullptr;
    if (mhip_check(hipHostMalloc(&p, bytes ? bytes : 256, hipHostMallocDefault), "hipHostMalloc")) return nullptr;
    return p;
}

extern "C" void mhip_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

extern "C" int mhip_memset_async(void *dst, int value, size_t bytes) {
    return mhip_check(hipMemsetAsync(dst, value, bytes, mhip_stream_native()), "hipMemsetAsync");
}
extern "C" int mhip_h2d_async(void *dst, const void *src, size_t bytes) {
    return mhip_check(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, mhip_stream_native()), "H2D");
}
extern "C" int mhip_d2h_async(void *dst, const void *src, size_t bytes) {
    return mhip_check(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, mhip_stream_native()), "D2H");
}
extern "C" int mhip_d2d_async(void *dst, const void *src, size_t bytes) {
    return mhip_check(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, mhip_stream_native()), "D2D");
}
extern "C" int mhip_h2d_2d_async(void *dst, size_t dpitch, const void *src, size_t spitch, size_t row_bytes,
                                 size_t rows) {
    if (dpitch == row_bytes && spitch == row_bytes) return mhip_h2d_async(dst, src, row_bytes * rows);
    return mhip_check(hipMemcpy2DAsync(dst, dpitch, src, spitch, row_bytes, rows, hipMemcpyHostToDevice, mhip_stream_native()), "H2D 2D");
}
extern "C" int mhip_d2h_2d_async(void *dst, size_t dpitch, const void *src, size_t spitch, size_t row_bytes,
                                 size_t rows) {
    if (dpitch == row_bytes && spitch == row_bytes) return mhip_d2h_async(dst, src, row_bytes * rows);
    return mhip_check(hipMemcpy2DAsync(dst, dpitch, src, spitch, row_bytes, rows, hipMemcpyDeviceToHost, mhip_stream_native()), "D2H 2D");
}

// ---- HIP graphs: a launch-bound plan (single frames: ~60 launches of a few microseconds each) is captured once from the
// main stream and replayed with one call.  Relaxed capture mode: the launchers' occupancy queries are not stream work.
extern "C" int mhip_graph_begin(void) {
    return mhip_check(hipStreamBeginCapture(g_stream, hipStreamCaptureModeRelaxed), "hipStreamBeginCapture");
}
extern "C" void *mhip_graph_end(int ok) { // ends the capture; ok == 0 (a launch failed): discard
    hipGraph_t graph = nullptr;
    if (hipStreamEndCapture(g_stream, &graph) != hipSuccess || !graph) return nullptr;
    hipGraphExec_t exec = nullptr;
    if (ok && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) exec = nullptr;
    (void)hipGraphDestroy(graph);
    return (void *)exec;
}
extern "C" int mhip_graph_launch(void *exec) {
    return mhip_check(hipGraphLaunch((hipGraphExec_t)exec, g_stream), "hipGraphLaunch");
}
extern "C" void mhip_graph_destroy(void *exec) {
    if (exec) (void)hipGraphExecDestroy((hipGraphExec_t)exec);
}

extern "C" void *mhip_event_create(void) {
    hipEvent_t ev;
    if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    return (void *)ev;
}
// ordering-only events (stream -> stream hand-offs): no timestamps, so recording and waiting stay on the device's queues
extern "C" void *mhip_event_create_sync(void) {
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return nullptr;
    return (void *)ev;
}
extern "C" void mhip_event_destroy(void *ev) {
    if (ev) (void)hipEventDestroy((hipEvent_t)ev);
}
extern "C" int mhip_event_record(void *ev) {
    return mhip_check(hipEventRecord((hipEvent_t)ev, mhip_stream_native()), "hipEventRecord");
}
extern "C" float mhip_event_elapsed_ms(void *start, void *stop) {
    float ms = 0.f;
    if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess) return -1.f;
    if (hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) return -1.f;
    return ms;
}

