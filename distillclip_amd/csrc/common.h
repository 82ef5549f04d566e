// Common device/host helpers for the distill-step kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#include "dclip.h"   // DCLIP_OK / DCLIP_EINVAL (bad shape, dtype, alignment -> ValueError in Python) / DCLIP_ELAUNCH (-> RuntimeError)

// thread-local last-error string (include/dclip.h: dclip_last_error_string)
void dclip_set_error(const char* fmt, ...);

#define DCLIP_REQUIRE(cond, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            dclip_set_error(__VA_ARGS__);         \
            return DCLIP_EINVAL;                  \
        }                                         \
    } while (0)

// launch trace (capi.cpp): open() returns false unless dclip_trace_begin() enabled tracing
bool dclip_trace_open(int kind, double flops, double bytes, void* stream, int* slot, int d0 = 0, int d1 = 0, int d2 = 0, int d3 = 0);
void dclip_trace_close(int slot, void* stream);
struct TraceScope {
    int slot = -1; void* st;
    TraceScope(int kind, double flops, double bytes, void* stream, int d0 = 0, int d1 = 0, int d2 = 0, int d3 = 0) : st(stream) {
        if (!dclip_trace_open(kind, flops, bytes, stream, &slot, d0, d1, d2, d3)) slot = -1;
    }
    ~TraceScope() { if (slot >= 0) dclip_trace_close(slot, st); }
};
#define DCLIP_TRACE_GEMM_NT 0
#define DCLIP_TRACE_GEMM_TN 1
#define DCLIP_TRACE_LAYERNORM 2
#define DCLIP_TRACE_LOSS 3
#define DCLIP_TRACE_ATTN 4
#define DCLIP_TRACE_LN_BWD 5

// a failed launch, in words (capi.cpp): names a second HIP runtime in the process or an invisible device where that is the cause
void dclip_explain_hip_error(const char* what, int hip_error, const char* hip_text);
static inline int dclip_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        dclip_explain_hip_error(what, (int)e, hipGetErrorString(e));
        return DCLIP_ELAUNCH;
    }
    return DCLIP_OK;
}

__device__ __forceinline__ float bf2f(bf16_t x) { return (float)x; }
__device__ __forceinline__ bf16_t f2bf(float x) { return (bf16_t)x; }   // v_cvt_pk_bf16_f32: RNE, NaN-preserving

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

__device__ __forceinline__ float quick_gelu_f(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * x)); }   // v_rcp_f32: 1 ulp
// exact-GELU pieces with a branch-free erf (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7, i.e. at the f32 rounding level of
// libdevice's erff but ~4x fewer instructions; the activations are stored as bf16 anyway).  phi = exp(-x^2/2) is shared
// between erf(x / sqrt2) and the density term of the derivative.
__device__ __forceinline__ float erf_tail(float ax_over_sqrt2, float phi) {      // 1 - erf(|x|/sqrt2), phi = exp(-x^2/2)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax_over_sqrt2, 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    return p * t * phi;
}
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float phi = __expf(-0.5f * x * x);
    const float tail = erf_tail(fabsf(x) * 0.70710678118654752f, phi);
    const float cdf = x >= 0.f ? 1.f - 0.5f * tail : 0.5f * tail;                // 0.5 (1 + erf(x / sqrt2))
    return x * cdf;
}
__device__ __forceinline__ float dgelu_erf_f(float x) {
    // d/dx [x Phi(x)] = Phi(x) + x phi(x) / sqrt(2 pi)
    const float phi = __expf(-0.5f * x * x);
    const float tail = erf_tail(fabsf(x) * 0.70710678118654752f, phi);
    const float cdf = x >= 0.f ? 1.f - 0.5f * tail : 0.5f * tail;
    return fmaf(x * phi, 0.3989422804014327f, cdf);
}

// gelu(x) and gelu'(x) together: the derivative costs two more instructions once the pieces of gelu are there, so the forward
// epilogue can hand the backward its factor instead of the pre-activation (DCLIP_ACT_GELU_SAVE / DCLIP_ACT_MULAUX).
__device__ __forceinline__ void gelu_erf_both_f(float x, float& g, float& dg) {
    const float phi = __expf(-0.5f * x * x);
    const float tail = erf_tail(fabsf(x) * 0.70710678118654752f, phi);
    const float cdf = x >= 0.f ? 1.f - 0.5f * tail : 0.5f * tail;
    g = x * cdf;
    dg = fmaf(x * phi, 0.3989422804014327f, cdf);
}

// gelu'(x) saved for the backward as 8-bit fixed point (DCLIP_ACT_GELU_SAVE writes it, DCLIP_ACT_MULAUX multiplies by it): the
// derivative of the exact GELU lies in [-0.1289, 1.1289]; code q = rint((g' - DG_LO) / DG_STEP) in 0..255, value DG_LO + q DG_STEP.
// |error| <= DG_STEP / 2 = 2.5e-3: on dz = dy g'(z) the same relative L2 error as a bf16 g' (DESIGN.md section 7), at half the bytes.
#define DCLIP_DG_LO (-0.13f)
#define DCLIP_DG_STEP (1.26f / 255.f)
__device__ __forceinline__ unsigned dg_pack4(float a, float b, float c, float d) {
    const float inv = 255.f / 1.26f, off = 0.13f * (255.f / 1.26f);
    unsigned r = 0;
    r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(fmaf(a, inv, off)), 0, r);
    r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(fmaf(b, inv, off)), 1, r);
    r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(fmaf(c, inv, off)), 2, r);
    r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(fmaf(d, inv, off)), 3, r);
    return r;
}
__device__ __forceinline__ float dg_unpack(unsigned w, int byte) {      // byte: compile-time constant 0..3
    return fmaf((float)((w >> (8 * byte)) & 0xffu), DCLIP_DG_STEP, DCLIP_DG_LO);
}

// XCD-aware, bijective remap of a linear workgroup id: blocks that share an XCD (same id % 8 under the
// observed round-robin placement) get a contiguous chunk of the tile space, so neighbouring tiles share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}
