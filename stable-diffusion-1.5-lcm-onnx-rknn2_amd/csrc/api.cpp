// Error reporting, device probe and hipGraph helpers of the C ABI (include/lcm_hip.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define LCM_OK 0
#define LCM_EINVAL (-1)
#define LCM_ENODEV (-2)

static thread_local char g_err[512] = "";

void lcm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* lcm_last_error(void) { return g_err; }
extern "C" int lcm_version(void) { return 100; }

extern "C" int lcm_device_info(int dev, char* arch_buf, int buf_len, int* cu_count, uint64_t* hbm_bytes) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { lcm_set_error("no HIP device"); return LCM_ENODEV; }
    if (dev < 0 || dev >= n) { lcm_set_error("device %d out of range (%d)", dev, n); return LCM_EINVAL; }
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) { lcm_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e)); return (int)e; }
    if (arch_buf && buf_len > 0) { strncpy(arch_buf, prop.gcnArchName, buf_len - 1); arch_buf[buf_len - 1] = 0; }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (uint64_t)prop.totalGlobalMem;
    return n;
}

extern "C" int lcm_graph_begin(void* stream) {
    hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { lcm_set_error("hipStreamBeginCapture: %s", hipGetErrorString(e)); return (int)e; }
    return LCM_OK;
}

extern "C" int lcm_graph_end(void* stream, void** graph_exec_out) {
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
    if (e != hipSuccess) { lcm_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return (int)e; }
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (e != hipSuccess) { lcm_set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return (int)e; }
    *graph_exec_out = (void*)ex;
    return LCM_OK;
}

extern "C" int lcm_graph_launch(void* graph_exec, void* stream) {
    hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
    if (e != hipSuccess) { lcm_set_error("hipGraphLaunch: %s", hipGetErrorString(e)); return (int)e; }
    return LCM_OK;
}

extern "C" int lcm_graph_destroy(void* graph_exec) {
    if (graph_exec) hipGraphExecDestroy((hipGraphExec_t)graph_exec);
    return LCM_OK;
}
