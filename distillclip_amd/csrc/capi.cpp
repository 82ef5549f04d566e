// Library-level entry points of libdistillclip_hip.so (include/dclip.h).
#include <stdarg.h>
#include <stdio.h>
#include "common.h"

static thread_local char g_err[1024] = "";

void dclip_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int dclip_version(void) { return 4; }

// ---------------------------------------------------------------------------------------------------------------
// Which HIP runtime is this library bound to, and is it the only one in the process?  (include/dclip.h, "Load order".)  PyTorch-ROCm
// ships a libamdhip64 of its own and asks for it by a different name than this library's DT_NEEDED entry, so a process that maps this
// library first and torch second holds two runtimes with separate device state; kernels of the second one then fail with HIP's
// "no ROCm-capable device is detected", which says nothing about the cause.
// ---------------------------------------------------------------------------------------------------------------
#include <string.h>
#include <string>
#include <vector>
static std::vector<std::string> mapped_hip_runtimes() {
    std::vector<std::string> out;
    FILE* f = fopen("/proc/self/maps", "r");
    if (!f) return out;
    char line[1024];
    while (fgets(line, sizeof(line), f)) {
        const char* p = strstr(line, "libamdhip64");
        if (!p) continue;
        const char* path = strchr(line, '/');
        if (!path) continue;
        std::string s(path);
        while (!s.empty() && (s.back() == '\n' || s.back() == ' ')) s.pop_back();
        bool seen = false;
        for (const auto& o : out) seen = seen || o == s;
        if (!seen) out.push_back(s);
    }
    fclose(f);
    return out;
}

void dclip_explain_hip_error(const char* what, int hip_error, const char* hip_text) {
    const std::vector<std::string> rt = mapped_hip_runtimes();
    if (rt.size() > 1) {
        std::string all;
        for (const auto& r : rt) all += (all.empty() ? "" : ", ") + r;
        dclip_set_error("%s: %s — %zu HIP runtimes are mapped in this process (%s): libdistillclip_hip.so was loaded before the one PyTorch ships; "
                        "import torch (or load its libamdhip64) BEFORE this library (include/dclip.h, Load order)", what, hip_text, rt.size(), all.c_str());
    } else if (hip_error == (int)hipErrorNoDevice || hip_error == (int)hipErrorInvalidDevice) {
        dclip_set_error("%s: %s — the HIP runtime %s sees no usable device (is a GPU visible to this process? HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES)",
                        what, hip_text, rt.empty() ? "(not found in /proc/self/maps)" : rt[0].c_str());
    } else {
        dclip_set_error("%s: %s", what, hip_text);
    }
}

extern "C" int dclip_runtime_check(void) {
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    const std::vector<std::string> rt = mapped_hip_runtimes();
    if (rt.size() > 1) { dclip_explain_hip_error("dclip_runtime_check", (int)e, e == hipSuccess ? "more than one HIP runtime" : hipGetErrorString(e)); return DCLIP_ELAUNCH; }
    if (e != hipSuccess || n <= 0) { dclip_explain_hip_error("dclip_runtime_check", (int)(e == hipSuccess ? hipErrorNoDevice : e), e == hipSuccess ? "no device" : hipGetErrorString(e)); return DCLIP_ELAUNCH; }
    return DCLIP_OK;
}
extern "C" const char* dclip_arch(void) { return "gfx950"; }
extern "C" const char* dclip_last_error_string(void) { return g_err; }

// ---------------------------------------------------------------------------------------------------------------
// launch trace (profiling only): HIP events around every traced launch, on the stream the kernel is launched on.
// ---------------------------------------------------------------------------------------------------------------
#include <mutex>
#include <vector>

namespace {
struct TraceRec { hipEvent_t a, b; int kind; double flops; double bytes; int dims[4]; };
std::mutex g_trace_mu;
bool g_trace_on = false;
std::vector<TraceRec> g_trace;
size_t g_trace_cap = 0;
}  // namespace

bool dclip_trace_open(int kind, double flops, double bytes, void* stream, int* slot, int d0, int d1, int d2, int d3) {
    std::lock_guard<std::mutex> lk(g_trace_mu);
    if (!g_trace_on || g_trace.size() >= g_trace_cap) return false;
    TraceRec r;
    r.kind = kind; r.flops = flops; r.bytes = bytes;
    r.dims[0] = d0; r.dims[1] = d1; r.dims[2] = d2; r.dims[3] = d3;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return false;
    (void)hipEventRecord(r.a, (hipStream_t)stream);
    g_trace.push_back(r);
    *slot = (int)g_trace.size() - 1;
    return true;
}

void dclip_trace_close(int slot, void* stream) {
    std::lock_guard<std::mutex> lk(g_trace_mu);
    if (slot >= 0 && slot < (int)g_trace.size()) (void)hipEventRecord(g_trace[slot].b, (hipStream_t)stream);
}

extern "C" int dclip_trace_begin(int64_t max_records) {
    std::lock_guard<std::mutex> lk(g_trace_mu);
    for (auto& r : g_trace) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_trace.clear();
    g_trace_cap = (size_t)(max_records > 0 ? max_records : 0);
    g_trace.reserve(g_trace_cap);
    g_trace_on = true;
    return DCLIP_OK;
}

// writes up to `cap` records as (kind, milliseconds, flops, bytes) and returns the number of records traced
extern "C" int64_t dclip_trace_end(int32_t* kind, float* ms, double* flops, double* bytes, int64_t cap) {
    std::lock_guard<std::mutex> lk(g_trace_mu);
    g_trace_on = false;
    int64_t n = 0;
    for (auto& r : g_trace) {
        (void)hipEventSynchronize(r.b);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, r.a, r.b);
        if (n < cap) { kind[n] = r.kind; ms[n] = t; flops[n] = r.flops; bytes[n] = r.bytes; }
        ++n;
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    g_trace.clear();
    return n;
}

// the problem sizes of the traced calls (4 ints per record: GEMM M, N, K, variant; other kinds rows, width, 0, 0); call BEFORE
// dclip_trace_end, which consumes the records
extern "C" int64_t dclip_trace_dims(int32_t* dims, int64_t cap) {
    std::lock_guard<std::mutex> lk(g_trace_mu);
    int64_t n = 0;
    for (auto& r : g_trace) {
        if (n < cap) for (int i = 0; i < 4; ++i) dims[n * 4 + i] = r.dims[i];
        ++n;
    }
    return n;
}
