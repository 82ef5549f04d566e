// gemm.hip (128^2 / 192- / 256- / 320-row kernels): the launch descriptor, the sub-tile swizzle and the per-unit epilogue (8 consecutive output columns of one row, straight from registers).
#pragma once
#include <math.h>
#include <stdlib.h>
#include "common.h"
#include <type_traits>

namespace dgemm {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int SUB = 1024;                       // bytes of one [16 x 32] bf16 sub-tile

struct GemmNT {
    const bf16_t* A; int64_t lda;
    const bf16_t* B; int64_t ldb;
    void* C; int64_t ldc;
    int M, N, K;
    float alpha;
    const float* bias;
    const void* aux_in;                             // ACT 3: bf16 pre-activation ; ACT 4: u8 fixed-point gelu' (dg_pack4)
    void* aux_out;                                  // ACT 5: u8 gelu' ; other ACT: bf16 pre-activation
    const void* residual; int64_t ldr;              // OUT 1: f32 ; OUT 2: f16 ; OUT 0: f32 (never aliases C)
    int row_group; const float* rowadd;
    float* colsum;                                  // += column sums of the epilogue output (bias gradient), may be null
    int tiles_m, tiles_n;
    unsigned long long* stamps;                     // profiling only (dclip_trace_gemm_stamps): 6 x u64 per workgroup, else null
    int group_n;                                    // 256-/320-row kernels: column tiles per raster group (>= tiles_n: plain n-fastest)
    unsigned long long* clk;                        // measurement only (dclip_trace_gemm_clock): 4 x u64 of this launch, else null
};

__device__ __forceinline__ int swz(int x) { return x ^ (((x >> 9) & 1) << 5); }

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// side operands of one output row segment, loaded ahead of use: residual and C may alias (in-place residual stream), so the
// compiler cannot hoist these loads above the previous row's store by itself — left inside the row loop every row pays a
// full HBM round trip in sequence
struct EpiSide {
    float4 r0, r1;      // residual (f32 stream)
    f16x8 rh;           // residual (f16 stream)
    bf16x8 z;           // aux_in, ACT 3
    uint2 zq;           // aux_in, ACT 4
};

// OUT: 0 bf16, 1 f32, 2 f16 output.  (The residual travels in EpiSide only with f32 / f16 output — the in-place residual stream; with
//  bf16 output it cannot alias C and is read inline.)
template <int ACT, int OUT>
__device__ __forceinline__ void epilogue_load_side(const GemmNT& p, int64_t o, int64_t orr, EpiSide& sd) {
    if (OUT == 1 && p.residual) {
        const float* rp = (const float*)p.residual + orr;
        sd.r0 = *(const float4*)rp; sd.r1 = *(const float4*)(rp + 4);
    }
    if (OUT == 2 && p.residual) sd.rh = *(const f16x8*)((const _Float16*)p.residual + orr);
    if (ACT == 3) sd.z = *(const bf16x8*)((const bf16_t*)p.aux_in + o);
    if (ACT == 4) sd.zq = *(const uint2*)((const uint8_t*)p.aux_in + o);
}

// MODE 0: every optional operand is a run-time test (wave-uniform branches: four per 8-column unit, 80 per 320 x 256 tile and wave —
// about 40 % of the issue slots of a plain bf16 epilogue, which is issue-bound).  MODE 1 / 2: the launch has no positional table, no
// residual and no saved pre-activation (ACT 5 / 6 always save their derivative), without / with bias-gradient column sums — the
// combinations the step's bf16 GEMMs use; the tests are compiled out.
template <int ACT, int OUT, int MODE = 0>
__device__ __forceinline__ void epilogue_vec8(const GemmNT& p, float (&v)[8], int row, int col, int64_t o, int64_t orr,
                                              const float (&bias)[8], float (&csum)[8], const EpiSide& sd) {
    constexpr bool LEAN = MODE != 0;            // MODE 3: f32 / f16 output with the (in-place) residual, nothing else optional
    // o = row * ldc + col, orr = row * ldr + col (formed incrementally by the caller)
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] * p.alpha + bias[e];
    if (!LEAN && p.row_group > 0) {      // (patch-embedding GEMM only: the position-embedding rows, L2-resident)
        const float* ra = p.rowadd + (int64_t)(row % p.row_group) * p.N + col;
        const float4 a0 = *(const float4*)ra, a1 = *(const float4*)(ra + 4);
        v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w; v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
    }
    if (!LEAN && ACT != 5 && ACT != 6 && p.aux_out) {
        bf16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = f2bf(v[e]);
        *(bf16x8*)((bf16_t*)p.aux_out + o) = z;
    }
    if (ACT == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = quick_gelu_f(v[e]);
    }
    if (ACT == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_erf_f(v[e]);
    }
    if (ACT == 3) {
        const bf16x8 z = sd.z;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= dgelu_erf_f(bf2f(z[e]));
    }
    if (ACT == 4) {
        const uint2 q = sd.zq;
        v[0] *= dg_unpack(q.x, 0); v[1] *= dg_unpack(q.x, 1); v[2] *= dg_unpack(q.x, 2); v[3] *= dg_unpack(q.x, 3);
        v[4] *= dg_unpack(q.y, 0); v[5] *= dg_unpack(q.y, 1); v[6] *= dg_unpack(q.y, 2); v[7] *= dg_unpack(q.y, 3);
    }
    if (ACT == 5) {
        float dg[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float g;
            gelu_erf_both_f(v[e], g, dg[e]);
            v[e] = g;
        }
        if (LEAN || p.aux_out) *(uint2*)((uint8_t*)p.aux_out + o) = uint2{dg_pack4(dg[0], dg[1], dg[2], dg[3]), dg_pack4(dg[4], dg[5], dg[6], dg[7])};
    }
    if (ACT == 6) {      // QuickGELU of a trainable CLIP tower: x s(1.702 x) and its derivative s + 1.702 x s (1 - s), in [-0.1, 1.1]
        float dg[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v[e]));
            dg[e] = fmaf(1.702f * v[e] * sg, 1.f - sg, sg);
            v[e] *= sg;
        }
        if (LEAN || p.aux_out) *(uint2*)((uint8_t*)p.aux_out + o) = uint2{dg_pack4(dg[0], dg[1], dg[2], dg[3]), dg_pack4(dg[4], dg[5], dg[6], dg[7])};
    }
    if (MODE == 3 || (!LEAN && p.residual)) {
        if (OUT == 2) {
            const f16x8 r = sd.rh;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
        } else {
            float4 r0, r1;
            if (OUT == 1) { r0 = sd.r0; r1 = sd.r1; }
            else {
                const float* rp = (const float*)p.residual + orr;
                r0 = *(const float4*)rp; r1 = *(const float4*)(rp + 4);
            }
            v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
        }
    }
    if (OUT == 1) {
        float* cp = (float*)p.C + o;
        *(float4*)cp = float4{v[0], v[1], v[2], v[3]};
        *(float4*)(cp + 4) = float4{v[4], v[5], v[6], v[7]};
    } else if (OUT == 2) {
        f16x8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = (_Float16)v[e];
        *(f16x8*)((_Float16*)p.C + o) = ov;
    } else {
        bf16x8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = f2bf(v[e]);
        *(bf16x8*)((bf16_t*)p.C + o) = ov;
    }
    if (OUT == 0 && (MODE == 2 || (MODE == 0 && p.colsum))) {
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[e] += v[e];
    }
}

#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

}  // namespace dgemm
