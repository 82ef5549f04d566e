// Shared by gemm.hip (128^2 / 256-row / 320-row kernels) and gemm_duo.hip (two co-resident 4-wave workgroups per CU): the launch
// descriptor, the sub-tile swizzle and the per-unit epilogue (8 consecutive output columns of one row, straight from registers).
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
    const bf16_t* aux_in;
    bf16_t* aux_out;
    const float* residual; int64_t ldr;
    int row_group; const float* rowadd;
    float* colsum;                                  // += column sums of the epilogue output (bias gradient), may be null
    int tiles_m, tiles_n;
    unsigned long long* stamps;                     // profiling only (dclip_trace_gemm_stamps): 6 x u64 per workgroup, else null
    int group_n;                                    // 256-/320-row kernels: column tiles per raster group (>= tiles_n: plain n-fastest)
    unsigned long long* clk;                        // measurement only (dclip_trace_gemm_clock): 4 x u64 of this launch, else null
    int duo_prio;                                   // gemm_duo.hip: how a workgroup picks its priority against its CU neighbour
    unsigned* tile_ctr;                             // persistent 256-/320-row launches: 8 ticket counters (one per XCD chunk, 128 B apart,
                                                    // zero between launches: the last ticket of a chunk resets its counter), else null
};

__device__ __forceinline__ int swz(int x) { return x ^ (((x >> 9) & 1) << 5); }

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// side operands of one output row segment, loaded ahead of use: residual and C may alias (in-place residual stream), so the
// compiler cannot hoist these loads above the previous row's store by itself — left inside the row loop every row pays a
// full HBM round trip in sequence
struct EpiSide {
    float4 r0, r1;      // residual
    bf16x8 z;           // aux_in
};

// (the residual travels in EpiSide only with f32 output — the in-place residual stream; with bf16 output it cannot alias C and
//  is read inline)
template <int ACT, bool OUT_F32>
__device__ __forceinline__ void epilogue_load_side(const GemmNT& p, int64_t o, int64_t orr, EpiSide& sd) {
    if (OUT_F32 && p.residual) {
        const float* rp = p.residual + orr;
        sd.r0 = *(const float4*)rp; sd.r1 = *(const float4*)(rp + 4);
    }
    if (ACT == 3 || ACT == 4) sd.z = *(const bf16x8*)(p.aux_in + o);
}

// MODE 0: every optional operand is a run-time test (wave-uniform branches: four per 8-column unit, 80 per 320 x 256 tile and wave —
// about 40 % of the issue slots of a plain bf16 epilogue, which is issue-bound).  MODE 1 / 2: the launch has no positional table, no
// residual and no saved pre-activation (ACT 5 always saves its derivative), without / with bias-gradient column sums — the
// combinations the step's bf16 GEMMs use; the tests are compiled out.
template <int ACT, bool OUT_F32, int MODE = 0>
__device__ __forceinline__ void epilogue_vec8(const GemmNT& p, float (&v)[8], int row, int col, int64_t o, int64_t orr,
                                              const float (&bias)[8], float (&csum)[8], const EpiSide& sd) {
    constexpr bool LEAN = MODE != 0;            // MODE 3: f32 output with the (in-place) residual, nothing else optional
    // o = row * ldc + col, orr = row * ldr + col (formed incrementally by the caller)
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] * p.alpha + bias[e];
    if (!LEAN && p.row_group > 0) {      // (patch-embedding GEMM only: the position-embedding rows, L2-resident)
        const float* ra = p.rowadd + (int64_t)(row % p.row_group) * p.N + col;
        const float4 a0 = *(const float4*)ra, a1 = *(const float4*)(ra + 4);
        v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w; v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
    }
    if (!LEAN && ACT != 5 && p.aux_out) {
        bf16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = f2bf(v[e]);
        *(bf16x8*)(p.aux_out + o) = z;
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
        const bf16x8 z = sd.z;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= bf2f(z[e]);
    }
    if (ACT == 5) {
        bf16x8 dz;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float g, dg;
            gelu_erf_both_f(v[e], g, dg);
            v[e] = g; dz[e] = f2bf(dg);
        }
        if (LEAN || p.aux_out) *(bf16x8*)(p.aux_out + o) = dz;
    }
    if (MODE == 3 || (!LEAN && p.residual)) {
        float4 r0, r1;
        if (OUT_F32) { r0 = sd.r0; r1 = sd.r1; }
        else {
            const float* rp = p.residual + orr;
            r0 = *(const float4*)rp; r1 = *(const float4*)(rp + 4);
        }
        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
    }
    if (OUT_F32) {
        float* cp = (float*)p.C + o;
        *(float4*)cp = float4{v[0], v[1], v[2], v[3]};
        *(float4*)(cp + 4) = float4{v[4], v[5], v[6], v[7]};
    } else {
        bf16x8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = f2bf(v[e]);
        *(bf16x8*)((bf16_t*)p.C + o) = ov;
    }
    if (!OUT_F32 && (MODE == 2 || (MODE == 0 && p.colsum))) {
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[e] += v[e];
    }
}

#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// gemm_duo.hip: co-resident 4-wave workgroups (two per CU), persistent over tiles.  Returns DCLIP_OK / an error when it took the
// launch, 1 when the shape is not eligible or the selection (DCLIP_GEMM_DUO) left it to the 8-wave kernels.
template <int ACT> int launch_nt_duo(GemmNT p, bool out_f32, hipStream_t st);

}  // namespace dgemm
