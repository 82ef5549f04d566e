// Library-level entry points of libdistillclip_hip.so (include/dclip.h).
#include <stdarg.h>
#include <stdio.h>
#include "common.h"

static thread_local char g_err[512] = "";

void dclip_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int dclip_version(void) { return 3; }
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
