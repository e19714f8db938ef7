// Host-side runtime pieces of libhawq_mi355: error reporting, hipGraph capture/replay,
// HIP-event timing.  No global mutable state except the thread-local error string.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace {
thread_local char g_err[512] = "";
__global__ void probe_kernel(int *out) { *out = 950; }
}  // namespace

void hawq_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *hawq_last_error(void) { return g_err; }
extern "C" int hawq_abi_version(void) { return HAWQ_ABI_VERSION; }

extern "C" int hawq_device_ok(void) {
    int *d = nullptr, h = 0;
    HAWQ_CHECK_HIP(hipMalloc(&d, sizeof(int)));
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(1), 0, 0, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess || h != 950) {
        hawq_set_error("hawq_device_ok: gfx950 probe kernel failed (%s)", hipGetErrorString(e));
        return 1;
    }
    return 0;
}

extern "C" int hawq_graph_begin(void *stream) {
    HAWQ_CHECK_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return 0;
}

extern "C" int hawq_graph_end(void *stream, void **graph_exec_out) {
    HAWQ_REQUIRE(graph_exec_out, "hawq_graph_end: null output");
    hipGraph_t graph = nullptr;
    HAWQ_CHECK_HIP(hipStreamEndCapture((hipStream_t)stream, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        hawq_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
        return 1;
    }
    *graph_exec_out = (void *)exec;
    return 0;
}

extern "C" int hawq_graph_launch(void *graph_exec, void *stream) {
    HAWQ_REQUIRE(graph_exec, "hawq_graph_launch: null graph");
    HAWQ_CHECK_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return 0;
}

extern "C" int hawq_graph_destroy(void *graph_exec) {
    if (graph_exec) HAWQ_CHECK_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return 0;
}

extern "C" int hawq_event_create(void **ev) {
    HAWQ_REQUIRE(ev, "hawq_event_create: null output");
    hipEvent_t e;
    HAWQ_CHECK_HIP(hipEventCreate(&e));
    *ev = (void *)e;
    return 0;
}
extern "C" int hawq_event_record(void *ev, void *stream) {
    HAWQ_CHECK_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return 0;
}
extern "C" int hawq_event_elapsed_ms(void *ev_start, void *ev_stop, float *ms) {
    HAWQ_REQUIRE(ms, "hawq_event_elapsed_ms: null output");
    HAWQ_CHECK_HIP(hipEventSynchronize((hipEvent_t)ev_stop));
    HAWQ_CHECK_HIP(hipEventElapsedTime(ms, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
    return 0;
}
extern "C" int hawq_event_destroy(void *ev) {
    if (ev) HAWQ_CHECK_HIP(hipEventDestroy((hipEvent_t)ev));
    return 0;
}
