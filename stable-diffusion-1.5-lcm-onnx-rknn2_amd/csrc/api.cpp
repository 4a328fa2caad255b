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

// A stream that belongs to the caller alone (not one of a framework's pooled, recycled handles): a pipeline lane keys its
// split-K workspace, its captured graphs and its launch order on it.
extern "C" int lcm_stream_create(void** stream_out) {
    if (!stream_out) { lcm_set_error("stream_create: null out pointer"); return LCM_EINVAL; }
    hipStream_t s = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) { lcm_set_error("hipStreamCreateWithFlags: %s", hipGetErrorString(e)); return (int)e; }
    *stream_out = (void*)s;
    return LCM_OK;
}

extern "C" int lcm_stream_destroy(void* stream) {
    if (!stream) return LCM_OK;
    hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) { lcm_set_error("hipStreamDestroy: %s", hipGetErrorString(e)); return (int)e; }
    return LCM_OK;
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

// ---------------------------------------------------------------------------------------------------------------
// Live per-launch timing of the MFMA kernels (bench.py's roofline leg): between lcm_profile_begin and
// lcm_profile_end every contraction / attention launch is bracketed by two HIP events recorded on its own launch
// stream -- around the MAIN kernel only (a split-K combine launch is outside the bracket), labelled with the
// kernel instantiation so the numbers can be checked against rocprofv3's per-kernel averages.
// ---------------------------------------------------------------------------------------------------------------
#include <string>
#include <vector>
struct ProfEntry { std::string name; hipEvent_t e0, e1; };
static std::vector<ProfEntry> g_prof;
static bool g_prof_on = false;
static size_t g_prof_cap = 0;

extern "C" int lcm_profile_begin(int max_launches) {
    if (g_prof_on) { lcm_set_error("profile already running"); return LCM_EINVAL; }
    g_prof.clear();
    g_prof.reserve(max_launches > 0 ? max_launches : 0);
    g_prof_cap = max_launches > 0 ? (size_t)max_launches : 0;
    g_prof_on = true;
    return LCM_OK;
}

void lcm_prof_start(const char* name, hipStream_t s) {
    if (!g_prof_on || g_prof.size() >= g_prof_cap) return;
    ProfEntry e; e.name = name;
    if (hipEventCreate(&e.e0) != hipSuccess || hipEventCreate(&e.e1) != hipSuccess) return;
    (void)hipEventRecord(e.e0, s);
    g_prof.push_back(e);
}

void lcm_prof_stop(hipStream_t s) {
    if (!g_prof_on || g_prof.empty()) return;
    (void)hipEventRecord(g_prof.back().e1, s);
}

// writes "name\tmilliseconds\n" per launch into out (NUL terminated); returns the number of launches or <0
extern "C" int lcm_profile_end(char* out, int64_t cap) {
    if (!g_prof_on) { lcm_set_error("profile not running"); return LCM_EINVAL; }
    g_prof_on = false;
    int64_t pos = 0;
    int n = 0;
    for (auto& e : g_prof) {
        float ms = -1.f;
        if (hipEventSynchronize(e.e1) == hipSuccess) (void)hipEventElapsedTime(&ms, e.e0, e.e1);
        (void)hipEventDestroy(e.e0);
        (void)hipEventDestroy(e.e1);
        char line[256];
        const int len = snprintf(line, sizeof(line), "%s\t%.6f\n", e.name.c_str(), ms);
        if (out && pos + len + 1 < cap) { memcpy(out + pos, line, len); pos += len; ++n; }
    }
    if (out && cap > 0) out[pos] = 0;
    g_prof.clear();
    return n;
}
