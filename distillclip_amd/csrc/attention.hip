// Attention kernels (gfx950): per-(batch, head) bf16 MFMA products + the head-mixing softmax stage.
//
//   reference teacher: model/component/_common.py:73-89   (QK^T / sqrt(hd), + causal mask, softmax, PV)
//   reference student: model/component/weight_share_model.py:101-125
//                      (q *= scale ; QK^T ; conv_l over heads ; softmax ; conv_w over heads ; PV)
//
// The sequences are tiny (N = 50 / 77 / 101) so one wave owns one (b, h) problem and the whole row of scores.
// Three MFMA product shapes cover forward and backward (all v_mfma_f32_16x16x32_bf16, f32 accumulate):
//   NT  C[i,j] = a * sum_d A[i,d] B[j,d]     scores S = QK^T ; dR = dO V^T        (fragments straight from HBM/L2)
//   NN  C[i,d] = a * sum_j A[i,j] B[j,d]     O = R V ; dQ = dS K                  (B through LDS + tr16 reads)
//   TN  C[j,d] = a * sum_i A[i,j] B[i,d]     dV = R^T dO ; dK = dS^T Q            (A and B through LDS + tr16 reads)
// The softmax / head-mix stage is fp32 VALU with one wave per (b, query row); its weight gradients
// (dW_l, dW_w: H x H, reduced over B*N*N positions) run on 32x32x16 MFMA from LDS tiles.
//
// Score-like tensors live as [B, H, N, Np] with Np = round_up(N, 8) (16-byte rows); pad columns are zero.
// q/k/v/ctx are token-major: row = b*N + n, column = head*hd + d (+ which*D inside the fused qkv buffer).
#include <stdlib.h>
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct AttnMM {
    const void* A; int64_t lda;      // NT: token-major bf16 ; NN/TN: [B,H,N,Np] bf16 (lda = Np)
    const bf16_t* Bm; int64_t ldb;   // token-major bf16
    void* C; int64_t ldc;            // NT: [B,H,N,Np] (f32 or bf16, ldc = Np) ; NN/TN: token-major bf16
    int B, H, N, Np, hd;
    float alpha;
    int a_blocked;                   // NN/TN: A is quad-blocked [B,H,Np/4,N,4] (attention_mix.hip) instead of row-major [B,H,N,Np]
};

// 8 consecutive columns j0 .. j0 + 7 (j0 % 8 == 0) of row i of a score-like matrix of one (b, h): row-major rows of Np, or the
// quad-blocked layout of the register-resident score stage (element (i, j) at ((j >> 2) * N + i) * 4 + (j & 3)): two 8-byte halves
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
__device__ __forceinline__ u32x4 load_a8(const bf16_t* A, int64_t lda, int N, int blocked, int i, int j0) {
    if (!blocked) return *(const u32x4*)(A + (int64_t)i * lda + j0);
    const u32x2 lo = *(const u32x2*)(A + ((int64_t)(j0 >> 2) * N + i) * 4);
    const u32x2 hi = *(const u32x2*)(A + ((int64_t)((j0 >> 2) + 1) * N + i) * 4);
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
}

__device__ __forceinline__ bf16x8 zero_frag() {
    bf16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = f2bf(0.f);
    return z;
}

// ---------------------------------------------------------------------------------------------------------
// NT: one wave per (b,h).  HD = head dim (32 or 64).
// ---------------------------------------------------------------------------------------------------------
template <int HD, bool OUT_F32>
__global__ __launch_bounds__(256) void attn_nt_kernel(AttnMM p) {
    const int lane = threadIdx.x & 63;
    const int prob = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (prob >= p.B * p.H) return;
    const int b = prob / p.H, h = prob % p.H;
    const bf16_t* A = (const bf16_t*)p.A + (int64_t)b * p.N * p.lda + h * HD;
    const bf16_t* Bm = p.Bm + (int64_t)b * p.N * p.ldb + h * HD;
    const int nt = (p.N + 15) >> 4;
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    constexpr int KS = HD / 32;
    constexpr int NTM = 8;                               // N <= 128
    // the B-side fragments (all key tiles) are the same for every query tile: load them once, keep them in registers
    bf16x8 bfr[NTM][KS];
#pragma unroll
    for (int jt = 0; jt < NTM; ++jt)
        if (jt < nt) {
            const int jb = min(jt * 16 + fr, p.N - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) bfr[jt][ks] = *(const bf16x8*)(Bm + (int64_t)jb * p.ldb + ks * 32 + fk);
        }
    // every query-tile fragment is requested before the first store: vmcnt retires loads and stores in one in-order queue, so a
    // load issued after a store cannot be waited for without waiting for that store's acknowledgement as well
    bf16x8 afa[NTM][KS];
#pragma unroll
    for (int it = 0; it < NTM; ++it)
        if (it < nt) {
            const int ia = min(it * 16 + fr, p.N - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) afa[it][ks] = *(const bf16x8*)(A + (int64_t)ia * p.lda + ks * 32 + fk);
        }
#pragma unroll
    for (int it = 0; it < NTM; ++it) {
        if (it < nt) {
#pragma unroll
            for (int jt = 0; jt < NTM; ++jt) {
                if (jt < nt) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afa[it][ks], bfr[jt][ks], acc, 0, 0, 0);
                    const int j = jt * 16 + fr;
                    if (j < p.Np) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = it * 16 + (lane >> 4) * 4 + r;
                            if (i < p.N) {
                                const float v = j < p.N ? acc[r] * p.alpha : 0.f;
                                const int64_t o = (((int64_t)b * p.H + h) * p.N + i) * p.ldc + j;
                                if (OUT_F32) ((float*)p.C)[o] = v;
                                else ((bf16_t*)p.C)[o] = f2bf(v);
                            }
                        }
                    }
                }
            }
        }
    }
}

// k-major LDS tile fragment (16 columns from x0, 32 rows from r0): two ds_read_b64_tr_b16
template <int ROWB>
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int r0, int x0, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const char* a0 = tile + (r0 + 8 * g + q) * ROWB + (x0 + 4 * pp) * 2;
    union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
    u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * ROWB));
    return u.v;
}

// B fragment of v_mfma_f32_32x32x16_bf16 from a k-major LDS tile (rows = contraction index, 32 columns from x0,
// 16 rows from r0): lane (col = l & 31, k = 8*(l >> 5) + e).  Each 16-lane group does two ds_read_b64_tr_b16.
template <int ROWB>
__device__ __forceinline__ bf16x8 tr_frag32(const char* tile, int r0, int x0, int lane) {
    const int g4 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const char* a0 = tile + (r0 + 8 * (g4 >> 1) + q) * ROWB + (x0 + 16 * (g4 & 1) + 4 * pp) * 2;
    union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
    u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * ROWB));
    return u.v;
}

__device__ __forceinline__ float lane_bcast(float v, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}

// copy `rows` x `cols` bf16 (cols % 8 == 0) from global (row stride ld) into a wave-private LDS tile (row stride ROWB
// bytes), zero-filling rows >= rows_valid ; rows is a multiple of 32
template <int ROWB>
__device__ __forceinline__ void wave_stage(const bf16_t* __restrict__ G, int64_t ld, int row0, int rows_valid, int rows,
                                           int cols, char* tile, int lane) {
    const int cpr = cols >> 3;                 // 16-byte chunks per row
    const int total = rows * cpr;
    for (int idx = lane; idx < total; idx += 64) {
        const int r = idx / cpr, c = idx - r * cpr;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row0 + r < rows_valid) v = *(const u32x4*)(G + (int64_t)(row0 + r) * ld + c * 8);
        *(u32x4*)(tile + r * ROWB + c * 16) = v;
    }
}

// wave_stage with the columns permuted in groups of four: group q = DT * g + dt of a row lands at group 4 * dt + g.  A transposing
// fragment read of column block dt then hands MFMA row 4 g + r the column (4 DT) g + 4 dt + r, so that in a product computed
// transposed (O^T = V^T P^T) lane (query c, group g) ends up with the 4 DT CONSECUTIVE output columns (4 DT) g .. + 4 DT - 1 of its
// query: 16- / 32-byte row-contiguous stores instead of 8-byte pieces on 64 different lines per instruction.
template <int ROWB, int DT>
__device__ __forceinline__ void wave_stage_perm4(const bf16_t* __restrict__ G, int64_t ld, int rows_valid, int rows, char* tile, int lane,
                                                 int row0 = 0) {
    // stages tile rows [row0, rows)
    constexpr int cpr = DT * 2;                // 16-byte chunks per row (HD = 16 DT)
    constexpr int BATCH = 8;                   // loads in flight per lane: a load -> write loop pays one HBM round trip per chunk
    const int total = (rows - row0) * cpr;
    for (int base = 0; base < total; base += BATCH * 64) {
        u32x4 v[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int idx = base + k * 64 + lane;
            const int r = row0 + idx / cpr, c = idx % cpr;
            v[k] = u32x4{0u, 0u, 0u, 0u};
            if (idx < total && r < rows_valid) v[k] = *(const u32x4*)(G + (int64_t)r * ld + c * 8);
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int idx = base + k * 64 + lane;
            const int r = row0 + idx / cpr, c = idx % cpr;
            const int q0 = 2 * c, q1 = 2 * c + 1;
            const int p0 = 4 * (q0 % DT) + q0 / DT, p1 = 4 * (q1 % DT) + q1 / DT;
            if (idx < total) {
                *(u32x2*)(tile + r * ROWB + p0 * 8) = u32x2{v[k][0], v[k][1]};
                *(u32x2*)(tile + r * ROWB + p1 * 8) = u32x2{v[k][2], v[k][3]};
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// NN: C[(b,i), h*HD + d] = alpha * sum_j A[b,h,i,j] * B[(b,j), h*HD + d]
// ---------------------------------------------------------------------------------------------------------
constexpr int NMAX = 128;                      // max padded sequence length handled by the attention kernels

template <int HD>
__global__ __launch_bounds__(256) void attn_nn_kernel(AttnMM p) {
    constexpr int ROWB = HD * 2 + 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nprob = p.B * p.H;
    const int prob = min(blockIdx.x * 4 + wave, nprob - 1);
    const bool live = blockIdx.x * 4 + wave < nprob;
    const int b = prob / p.H, h = prob % p.H;
    const int n32 = (p.N + 31) & ~31;
    char* tile = smem + wave * (n32 * ROWB);        // sized by the padded sequence, not by NMAX: N = 50 fits 6 workgroups per CU instead of 3
    wave_stage_perm4<ROWB, HD / 16>(p.Bm + (int64_t)b * p.N * p.ldb + h * HD, p.ldb, p.N, n32, tile, lane);
    __syncthreads();
    const bf16_t* A = (const bf16_t*)p.A + ((int64_t)b * p.H + h) * p.N * p.lda;
    const int nt = (p.N + 15) >> 4, nks = n32 >> 5;
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    constexpr int DT = HD / 16;
    constexpr int KSM = NMAX / 32;
    bf16x8 af[KSM], afn[KSM];
    auto load_a = [&](int it, bf16x8 (&dst)[KSM]) {
        const int ia = min(it * 16 + fr, p.N - 1);
#pragma unroll
        for (int ks = 0; ks < KSM; ++ks) {
            const int j0 = ks * 32 + fk;
            dst[ks] = (ks < nks && j0 < p.Np) ? __builtin_bit_cast(bf16x8, load_a8(A, p.lda, p.N, p.a_blocked, ia, j0)) : zero_frag();
        }
    };
    load_a(0, af);
    for (int it = 0; it < nt; ++it) {
        if (it + 1 < nt) load_a(it + 1, afn);
        f32x4 acc[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) acc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSM; ++ks) {
            if (ks < nks) {
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    // computed transposed (C^T = B^T A^T; both fragment kinds share one lane layout, so the operands just swap)
                    const bf16x8 bf = tr_frag<ROWB>(tile, ks * 32, d * 16, lane);
                    acc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf, af[ks], acc[d], 0, 0, 0);
                }
            }
        }
        const int i = it * 16 + fr;
        if (live && i < p.N) {
            // lane (query fr, group g) holds the 4 DT consecutive columns from (4 DT) g (wave_stage_perm4): 16-byte row-contiguous stores
            bf16_t* o = (bf16_t*)p.C + ((int64_t)b * p.N + i) * p.ldc + h * HD + (lane >> 4) * (4 * DT);
#pragma unroll
            for (int d = 0; d < DT; d += 2)
                *(bf16x8*)(o + d * 4) = bf16x8{f2bf(acc[d][0] * p.alpha), f2bf(acc[d][1] * p.alpha), f2bf(acc[d][2] * p.alpha), f2bf(acc[d][3] * p.alpha),
                                                f2bf(acc[d + 1][0] * p.alpha), f2bf(acc[d + 1][1] * p.alpha), f2bf(acc[d + 1][2] * p.alpha), f2bf(acc[d + 1][3] * p.alpha)};
        }
#pragma unroll
        for (int ks = 0; ks < KSM; ++ks) af[ks] = afn[ks];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Fused multi-head attention forward (no head mixing: the frozen CLIP teacher, reference _common.py:73-89):
//   ctx[(b,i), h*HD + d] = sum_j softmax_j(scale * q_i . k_j (+ causal mask)) v_j[d]
// One wave per (b, h).  Everything is computed transposed so that the query index sits on the lane:
//   S^T tile = mfma(K frag, Q frag)  -> a lane holds 4 consecutive keys of ONE query: softmax needs 2 cross-lane steps,
//   P goes to a wave-private LDS tile with 8-byte writes, O^T = V^T P^T (V through LDS + tr16 reads) -> 8-byte stores.
// Scores and probabilities never touch HBM.
// ---------------------------------------------------------------------------------------------------------
struct AttnFused {
    const bf16_t* qkv; int64_t ldq;      // [B*N, 3*H*HD] : q | k | v
    bf16_t* ctx; int64_t ldc;
    int B, H, N, causal;
    float scale;
    int split;                           // waves per (b, h) problem: 1, or 2 that share the V tile and take alternate query tiles
};

// NTM = 16-key tiles the instance holds registers for (4: N <= 64, 5: N <= 80, 7: N <= 112, 8: N <= 128); the K fragments and score tiles
// scale with it, and at NTM = 4 / 5 three waves per SIMD fit where the N = 128 sizing allowed two.
// LDS (round 5): the V and P tiles hold 16 x ceil(N / 16) key rows, not the sequence padded to 32 — an odd number of key tiles ends in
// ONE v_mfma_f32_16x16x16_bf16 step instead of a half-empty 32-key step —, and the launch picks the workgroup size (4, 2 or 1 waves) that
// puts the most waves on a CU: at N = 101 (l_clip at 336 px) a wave needs 21.8 KB instead of 24.8 and runs in one-wave workgroups, 7 per
// CU where the four-wave workgroup of 99 KB left ONE per CU (151 us for the ViT-B/32 teacher against 38.7 at N = 50: 3.9 x for 2 x tokens).
template <int HD, int NTM>
__global__ __launch_bounds__(256, NTM <= 7 ? 3 : 2) void attn_fused_fwd_kernel(AttnFused p) {
    constexpr int VROWB = HD * 2 + 32;
    constexpr int KS = HD / 32, DT = HD / 16, KSM = (NTM + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6, split = p.split, ppw = nw / split;      // ppw: problems per workgroup
    const int nprob = p.B * p.H;
    const int slot = wave / split, part = wave - slot * split;
    const int prob = min((int)blockIdx.x * ppw + slot, nprob - 1);
    const bool live = (int)blockIdx.x * ppw + slot < nprob;
    const int b = prob / p.H, h = prob % p.H;
    const int D = p.H * HD;
    constexpr bool ODD = (NTM & 1) != 0;              // odd instances serve exactly NTM key tiles: whole 32-key steps + one 16-key step
    const int nt = (p.N + 15) >> 4;
    const int nrows = ODD ? nt * 16 : ((nt + 1) & ~1) * 16, nks = ODD ? nt >> 1 : (nt + 1) >> 1;
    const int prowb = nrows * 2 + 16;
    // LDS: [problems per workgroup] V tiles, then [waves] P tiles
    char* vt = smem + slot * (nrows * VROWB);
    char* pt = smem + ppw * (nrows * VROWB) + wave * (16 * prowb);
    const bf16_t* Q = p.qkv + (int64_t)b * p.N * p.ldq + h * HD;
    const bf16_t* K = Q + D;
    const bf16_t* V = Q + 2 * D;
    const int fr = lane & 15, g = lane >> 4, fk = g * 8;
    bf16x8 kf[NTM][KS];
#pragma unroll
    for (int jt = 0; jt < NTM; ++jt)
        if (jt < nt) {
            const int jb = min(jt * 16 + fr, p.N - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kf[jt][ks] = *(const bf16x8*)(K + (int64_t)jb * p.ldq + ks * 32 + fk);
        }
    bf16x8 qf[KS], qn[KS];
    {
        const int ia = min(part * 16 + fr, p.N - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(Q + (int64_t)ia * p.ldq + ks * 32 + fk);
    }
    // V last: its loads join the K / Q requests already in flight, and only its LDS writes wait
    {   // the waves of a problem stage disjoint row ranges of its V tile
        const int per = ((nrows / split) + 15) & ~15;
        wave_stage_perm4<VROWB, DT>(V, p.ldq, p.N, min(nrows, (part + 1) * per), vt, lane, part * per);
    }
    for (int idx = lane; idx < 16 * prowb / 4; idx += 64) ((unsigned*)pt)[idx] = 0u;
    __syncthreads();
    for (int it = part; it < nt; it += split) {
        if (it + split < nt) {
            const int ia = min((it + split) * 16 + fr, p.N - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qn[ks] = *(const bf16x8*)(Q + (int64_t)ia * p.ldq + ks * 32 + fk);
        }
        const int i = it * 16 + fr;                         // this lane's query
        // S^T tiles: acc[jt][r] = raw score(query i, key jt*16 + 4g + r); the scale rides in the exponent's fma (scale > 0: the maximum of
        // the raw scores is the maximum of the scaled ones), masked keys are -inf and come out of exp2 as 0, and the probabilities are
        // stored UNNORMALISED (e in [0, 1]: the same relative bf16 precision) with 1 / sum applied to the 4 DT outputs instead of to
        // every one of the N probabilities: 4.5 vector instructions per score element where the first version spent ~10
        f32x4 st[NTM];
        float m = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < NTM; ++jt) {
            st[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (jt < nt && (!p.causal || jt <= it)) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) st[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[jt][ks], qf[ks], st[jt], 0, 0, 0);
                if (jt * 16 + 16 > p.N || (p.causal && jt == it)) {        // (wave-uniform: only the last key tile / the diagonal tile mask)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = jt * 16 + g * 4 + r;
                        const bool ok = j < p.N && (!p.causal || j <= i);
                        st[jt][r] = ok ? st[jt][r] : -INFINITY;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, st[jt][r]);
            } else {
                st[jt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            }
        }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        const float c2 = p.scale * 1.4426950408889634f, nm = -m * c2;      // (key 0 is never masked: m is finite)
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < NTM; ++jt)
            if (jt < nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(st[jt][r], c2, nm));
                    st[jt][r] = e;
                    sum += e;
                }
                *(bf16x4*)(pt + fr * prowb + (jt * 16 + g * 4) * 2) = bf16x4{f2bf(st[jt][0]), f2bf(st[jt][1]), f2bf(st[jt][2]), f2bf(st[jt][3])};
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.f / sum;
        // wave-private tile, in-order DS queue: only the compiler has to keep the order (a workgroup-scope release fence would also
        // wait for vmcnt(0), i.e. for the next tile's query fragments that were requested at the top of the iteration)
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // O^T[d, i] = sum_j V[j, d] P[i, j]
        f32x4 oc[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) oc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSM; ++ks)
            if (ks < nks && (!p.causal || ks * 32 <= it * 16 + 15)) {
                const bf16x8 pf = *(const bf16x8*)(pt + fr * prowb + (ks * 32 + fk) * 2);
#pragma unroll
                for (int d = 0; d < DT; ++d)
                    oc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<VROWB>(vt, ks * 32, d * 16, lane), pf, oc[d], 0, 0, 0);
            }
        if (ODD && (nt & 1) && (!p.causal || (nt - 1) * 16 <= it * 16 + 15)) {
            // the last, odd key tile: 16 keys, lane (column, g) holds k = 4 g .. 4 g + 3 of both operands
            const int r0 = (nt - 1) * 16;
            const s16x4 pf = *(const s16x4*)(pt + fr * prowb + (r0 + g * 4) * 2);
            const int q4 = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const s16x4 vf = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vt + (r0 + 4 * g + q4) * VROWB + (d * 16 + 4 * pp) * 2));
                oc[d] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf, pf, oc[d], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (live && i < p.N) {
            // (columns permuted when V was staged: this lane holds the 4 DT consecutive columns from (4 DT) g of query i)
            bf16_t* o = p.ctx + ((int64_t)b * p.N + i) * p.ldc + h * HD + g * (4 * DT);
#pragma unroll
            for (int d = 0; d < DT; d += 2)
                *(bf16x8*)(o + d * 4) = bf16x8{f2bf(oc[d][0] * inv), f2bf(oc[d][1] * inv), f2bf(oc[d][2] * inv), f2bf(oc[d][3] * inv),
                                                f2bf(oc[d + 1][0] * inv), f2bf(oc[d + 1][1] * inv), f2bf(oc[d + 1][2] * inv), f2bf(oc[d + 1][3] * inv)};
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = qn[ks];
    }
}

// ---------------------------------------------------------------------------------------------------------
// TN: C[(b,j), h*HD + d] = alpha * sum_i A[b,h,i,j] * B[(b,i), h*HD + d]   (contraction over query rows)
// ---------------------------------------------------------------------------------------------------------

// MAXJ = output row tiles (16 rows each) the instance holds accumulators for: 4 (N <= 64), 5 (N <= 80) or 8 (N <= 128)
template <int HD, int MAXJ>
__global__ __launch_bounds__(256, MAXJ <= 5 ? 2 : 1) void attn_tn_kernel(AttnMM p) {
    constexpr int BROW = HD * 2 + 32;
    constexpr int DT = HD / 16;
    constexpr int TN_MAXJ = MAXJ;
    constexpr int AROW = MAXJ * 32 + 32;            // LDS row of the A chunk [32 x 16 MAXJ] bf16, padded
    constexpr int ACH = 32 * (MAXJ * 2) / 64, BCH = 32 * (HD / 8) / 64;     // 16-byte chunks per lane of a 32-row chunk (A: Np <= 16 MAXJ columns)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nprob = p.B * p.H;
    const int prob = min(blockIdx.x * 4 + wave, nprob - 1);
    const bool live = blockIdx.x * 4 + wave < nprob;
    const int b = prob / p.H, h = prob % p.H;
    char* at = smem + wave * (32 * AROW + 32 * BROW);
    char* bt = at + 32 * AROW;
    const bf16_t* A = (const bf16_t*)p.A + ((int64_t)b * p.H + h) * p.N * p.lda;
    const bf16_t* Bm = p.Bm + (int64_t)b * p.N * p.ldb + h * HD;
    const int ntj = (p.N + 15) >> 4;
    const int nchunk = (p.N + 31) >> 5;
    const int acpr = p.Np >> 3;                       // A: 16-byte chunks per row
    f32x4 acc[TN_MAXJ][DT];
#pragma unroll
    for (int j = 0; j < TN_MAXJ; ++j)
#pragma unroll
        for (int d = 0; d < DT; ++d) acc[j][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    // register-staged pipeline (guide T14): chunk c+1 is loaded into registers before chunk c is consumed, and written to
    // the wave-private LDS tiles afterwards.  Rows >= N are zero (they must not contribute to the contraction).
    u32x4 ra[ACH], rb[BCH];
    auto load_chunk = [&](int ch) {
#pragma unroll
        for (int k = 0; k < ACH; ++k) {
            // lane -> (row r, 16-byte column chunk c): columns fastest for row-major A, ROWS fastest for the quad-blocked layout, whose
            // 8-byte pieces of consecutive rows are adjacent in memory (16 lanes = 128 contiguous bytes; with the columns fastest every
            // lane of a load touched a line of its own and the texture addresser, not HBM, set the kernel's time)
            const int idx = k * 64 + lane;
            const int r = p.a_blocked ? (idx & 31) : idx / acpr, c = p.a_blocked ? (idx >> 5) : idx - r * acpr;
            ra[k] = u32x4{0u, 0u, 0u, 0u};
            if (r < 32 && c < acpr && ch * 32 + r < p.N) ra[k] = load_a8(A, p.lda, p.N, p.a_blocked, ch * 32 + r, c * 8);
        }
#pragma unroll
        for (int k = 0; k < BCH; ++k) {
            const int idx = k * 64 + lane, r = idx / (HD / 8), c = idx % (HD / 8);
            rb[k] = u32x4{0u, 0u, 0u, 0u};
            if (ch * 32 + r < p.N) rb[k] = *(const u32x4*)(Bm + (int64_t)(ch * 32 + r) * p.ldb + c * 8);
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int k = 0; k < ACH; ++k) {
            const int idx = k * 64 + lane;
            const int r = p.a_blocked ? (idx & 31) : idx / acpr, c = p.a_blocked ? (idx >> 5) : idx - r * acpr;
            if (r < 32 && c < acpr) *(u32x4*)(at + r * AROW + c * 16) = ra[k];
        }
#pragma unroll
        for (int k = 0; k < BCH; ++k) {
            // columns permuted in groups of four as in wave_stage_perm4: the product is computed transposed and a lane ends up with
            // 4 DT consecutive output columns
            const int idx = k * 64 + lane, r = idx / (HD / 8), c = idx % (HD / 8);
            const int q0 = 2 * c, q1 = 2 * c + 1;
            *(u32x2*)(bt + r * BROW + (4 * (q0 % DT) + q0 / DT) * 8) = u32x2{rb[k][0], rb[k][1]};
            *(u32x2*)(bt + r * BROW + (4 * (q1 % DT) + q1 / DT) * 8) = u32x2{rb[k][2], rb[k][3]};
        }
    };
    load_chunk(0);
    for (int ch = 0; ch < nchunk; ++ch) {
        __builtin_amdgcn_wave_barrier();
        store_chunk();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (ch + 1 < nchunk) load_chunk(ch + 1);
        bf16x8 bf[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) bf[d] = tr_frag<BROW>(bt, 0, d * 16, lane);
#pragma unroll
        for (int j = 0; j < TN_MAXJ; ++j) {
            if (j < ntj) {
                // columns beyond Np were never staged: they only feed output rows >= N, which are not stored
                const bf16x8 af = tr_frag<AROW>(at, 0, j * 16, lane);
#pragma unroll
                for (int d = 0; d < DT; ++d)
                    acc[j][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[d], af, acc[j][d], 0, 0, 0);      // transposed: lane = output row
            }
        }
    }
    if (live) {
        bf16_t* C = (bf16_t*)p.C + (int64_t)b * p.N * p.ldc + h * HD + (lane >> 4) * (4 * DT);
#pragma unroll
        for (int j = 0; j < TN_MAXJ; ++j)
            if (j < ntj) {
                const int jj = j * 16 + (lane & 15);
                if (jj < p.N) {
                    bf16_t* o = C + (int64_t)jj * p.ldc;
#pragma unroll
                    for (int d = 0; d < DT; d += 2)
                        *(bf16x8*)(o + d * 4) = bf16x8{f2bf(acc[j][d][0] * p.alpha), f2bf(acc[j][d][1] * p.alpha), f2bf(acc[j][d][2] * p.alpha), f2bf(acc[j][d][3] * p.alpha),
                                                        f2bf(acc[j][d + 1][0] * p.alpha), f2bf(acc[j][d + 1][1] * p.alpha), f2bf(acc[j][d + 1][2] * p.alpha), f2bf(acc[j][d + 1][3] * p.alpha)};
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
// softmax with optional cross-head mixes, one wave per (b, query row i); lane <-> key j (+64 per slot)
//   A_g = sum_h Wl[g,h] S_h ; P_g = softmax_j(A_g) (causal: j <= i) ; R_g = sum_h Ww[g,h] P_h
// ---------------------------------------------------------------------------------------------------------
struct SoftmaxFwd {
    const float* S;          // [B,H,N,Np] f32
    const float* Wl;         // [H,H] or null
    const float* Ww;         // [H,H] or null
    bf16_t* P;               // [B,H,N,Np] or null (saved for backward when Ww is set)
    bf16_t* R;               // [B,H,N,Np]
    int B, N, Np, causal;
    int H;                   // (run-time copy: the kernel without head mixing takes any head count)
};

// plain multi-head softmax (no head mixing: a CLIP tower that trains), any head count: one wave per (b, query row), heads in turn
template <int NS>
__global__ __launch_bounds__(256) void attn_softmax_fwd_plain_kernel(SoftmaxFwd p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.B * p.N) return;
    const int b = row / p.N, i = row % p.N;
    const int64_t hs = (int64_t)p.N * p.Np;
    const int64_t base = ((int64_t)b * p.H * p.N + i) * p.Np;
    float nxt[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) nxt[s] = lane + 64 * s < p.N ? p.S[base + lane + 64 * s] : 0.f;
    for (int h = 0; h < p.H; ++h) {
        float a[NS];
        float m = -INFINITY;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int j = lane + 64 * s;
            a[s] = (j < p.N && (!p.causal || j <= i)) ? nxt[s] : -INFINITY;
            m = fmaxf(m, a[s]);
            if (h + 1 < p.H) nxt[s] = j < p.N ? p.S[base + (h + 1) * hs + j] : 0.f;       // the next head's row is in flight during this one's reductions
        }
        m = wave_max(m);
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            a[s] = a[s] == -INFINITY ? 0.f : __expf(a[s] - m);
            sum += a[s];
        }
        const float inv = 1.f / wave_sum(sum);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int j = lane + 64 * s;
            if (j < p.Np) {
                const bf16_t v = f2bf(a[s] * inv);
                if (p.P) p.P[base + h * hs + j] = v;
                p.R[base + h * hs + j] = v;
            }
        }
    }
}

template <int H, int NS>
__global__ __launch_bounds__(256) void attn_softmax_fwd_kernel(SoftmaxFwd p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.B * p.N) return;
    const int b = row / p.N, i = row % p.N;
    const int64_t hs = (int64_t)p.N * p.Np;                       // head stride
    const int64_t base = ((int64_t)b * H * p.N + i) * p.Np;
    float sv[NS][H];
    bool valid[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int j = lane + 64 * s;
        valid[s] = j < p.N && (!p.causal || j <= i);
#pragma unroll
        for (int h = 0; h < H; ++h) sv[s][h] = j < p.N ? p.S[base + h * hs + j] : 0.f;
    }
    float pr[NS][H];
#pragma unroll
    for (int g = 0; g < H; ++g) {
        float a[NS];
        float m = -INFINITY;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (p.Wl) {
                float t = 0.f;
#pragma unroll
                for (int h = 0; h < H; ++h) t = fmaf(p.Wl[g * H + h], sv[s][h], t);
                a[s] = t;
            } else {
                a[s] = sv[s][g];
            }
            if (!valid[s]) a[s] = -INFINITY;
            m = fmaxf(m, a[s]);
        }
        m = wave_max(m);
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            a[s] = valid[s] ? __expf(a[s] - m) : 0.f;
            sum += a[s];
        }
        const float inv = 1.f / wave_sum(sum);
#pragma unroll
        for (int s = 0; s < NS; ++s) pr[s][g] = a[s] * inv;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int j = lane + 64 * s;
        if (j < p.Np) {
#pragma unroll
            for (int g = 0; g < H; ++g) {
                if (p.P) p.P[base + g * hs + j] = f2bf(pr[s][g]);
                float r = pr[s][g];
                if (p.Ww) {
                    r = 0.f;
#pragma unroll
                    for (int h = 0; h < H; ++h) r = fmaf(p.Ww[g * H + h], pr[s][h], r);
                }
                p.R[base + g * hs + j] = f2bf(r);
            }
        }
    }
}

// Head-mixing softmax forward on MFMA (one wave per (b, query row), tiles [32 heads][COLS keys] in wave-private LDS):
//   A_g = sum_h Wl[g,h] S_h     3 MFMAs per step with split-bf16 operands (Wl_hi S_hi + Wl_hi S_lo + Wl_lo S_hi): ~16 mantissa
//                               bits on the pre-softmax scores instead of 8
//   e   = exp(A - rowmax)       ONE cross-lane max per query row (shared by the heads; softmax is shift-invariant per head)
//   sum_g = sum_j e[g,j]        MFMA of the e tile against a ones operand -> lands in accumulator layout (row g)
//   P = e / sum (saved for backward) ; R_g = sum_h Ww[g,h] P_h (MFMA) ; P and R leave through LDS as 16-byte rows
template <int H, int NS>
__global__ __launch_bounds__(256) void attn_softmax_fwd_mix_kernel(SoftmaxFwd p) {
    constexpr int COLS = 64 * NS, ROWB = COLS * 2 + 16, NCT = COLS / 32, TILE = 32 * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* tH = smem + wave * 3 * TILE;          // S_hi, later R
    char* tL = tH + TILE;                       // S_lo
    char* tP = tL + TILE;                       // e, then P
    for (int idx = lane; idx < 3 * TILE / 16; idx += 64) ((u32x4*)tH)[idx] = u32x4{0u, 0u, 0u, 0u};
    const int hh = lane >> 5, c = lane & 31;
    bf16x8 aLh[2], aLl[2], aWh[2], aWl2[2], ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = f2bf(1.f);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int h = 16 * s + 8 * hh + e;                                   // A[g = c][k = h]
            const float wl = (h < H && c < H) ? p.Wl[c * H + h] : 0.f;
            const bf16_t hi = f2bf(wl);
            aLh[s][e] = hi;
            aLl[s][e] = f2bf(wl - bf2f(hi));
            const float ww = (h < H && c < H) ? p.Ww[c * H + h] : 0.f;
            const bf16_t whi = f2bf(ww);
            aWh[s][e] = whi;
            aWl2[s][e] = f2bf(ww - bf2f(whi));
        }
    const int64_t hs = (int64_t)p.N * p.Np;
    const int rows = p.B * p.N;
    const int nchunk = p.Np >> 3, total = H * nchunk;
    // the next row's scores are fetched into registers BEFORE this row's P / R stores are issued: vmcnt retires loads and stores
    // in one in-order queue, so loads that follow the stores could only be waited for together with the stores' acknowledgements
    constexpr int NIT = (H * (COLS / 8) + 63) / 64;
    float4 q0[NIT], q1[NIT];
    auto fetch = [&](int row) {
        const int64_t fb = ((int64_t)(row / p.N) * H * p.N + row % p.N) * p.Np;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = lane + it * 64;
            if (idx < total) {
                const int h = idx / nchunk, ck = idx - h * nchunk;
                const int64_t src = fb + h * hs + ck * 8;
                q0[it] = *(const float4*)(p.S + src); q1[it] = *(const float4*)(p.S + src + 4);
            }
        }
    };
    const int row_first = blockIdx.x * 4 + wave, row_step = gridDim.x * 4;
    if (row_first < rows) fetch(row_first);
    for (int row = row_first; row < rows; row += row_step) {
        const int b = row / p.N, i = row % p.N;
        const int64_t base = ((int64_t)b * H * p.N + i) * p.Np;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = lane + it * 64;
            if (idx < total) {
                const int h = idx / nchunk, ck = idx - h * nchunk;
                const float4 s0 = q0[it], s1 = q1[it];
                const float v[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                bf16x8 hi, lo;
#pragma unroll
                for (int e = 0; e < 8; ++e) { hi[e] = f2bf(v[e]); lo[e] = f2bf(v[e] - bf2f(hi[e])); }
                *(bf16x8*)(tH + h * ROWB + ck * 16) = hi;
                *(bf16x8*)(tL + h * ROWB + ck * 16) = lo;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        f32x16 am[NCT];
        float m = -INFINITY;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            am[ct] = f32x16{0};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 bh = tr_frag32<ROWB>(tH, 16 * s, 32 * ct, lane), bl = tr_frag32<ROWB>(tL, 16 * s, 32 * ct, lane);
                am[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aLh[s], bh, am[ct], 0, 0, 0);
                am[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aLh[s], bl, am[ct], 0, 0, 0);
                am[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aLl[s], bh, am[ct], 0, 0, 0);
            }
            const bool jok = 32 * ct + c < p.N;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int g = (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (jok && g < H) m = fmaxf(m, am[ct][r]);
            }
        }
        m = wave_max(m);
        // e = exp(A - m) as bf16 rows [g][j] in LDS (pad keys / pad heads stay zero)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const bool jok = 32 * ct + c < p.N;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int g = (r & 3) + 8 * (r >> 2) + 4 * hh;
                const float e = (jok && g < H) ? __expf(am[ct][r] - m) : 0.f;
                am[ct][r] = e;
                *(bf16_t*)(tP + g * ROWB + (32 * ct + c) * 2) = f2bf(e);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // row sums on the matrix pipe: sum_g = sum_j e[g, j] * 1  (accumulator layout: row g in the registers)
        f32x16 rs = {0};
#pragma unroll
        for (int ks = 0; ks < COLS / 16; ++ks)
            rs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(tP + c * ROWB + (ks * 16 + hh * 8) * 2), ones, rs, 0, 0, 0);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) rs[r] = __builtin_amdgcn_rcpf(rs[r]);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int g = (r & 3) + 8 * (r >> 2) + 4 * hh;
                const float pv = g < H ? am[ct][r] * rs[r] : 0.f;
                const bf16_t phi = f2bf(pv);
                *(bf16_t*)(tP + g * ROWB + (32 * ct + c) * 2) = phi;                       // saved P (bf16) = the hi part
                *(bf16_t*)(tL + g * ROWB + (32 * ct + c) * 2) = f2bf(pv - bf2f(phi));    // lo part, over the dead S_lo tile
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // R = Ww P, written over the (dead) S_hi tile as bf16 rows
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            f32x16 rr = {0};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 ph = tr_frag32<ROWB>(tP, 16 * s, 32 * ct, lane), pl = tr_frag32<ROWB>(tL, 16 * s, 32 * ct, lane);
                rr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aWh[s], ph, rr, 0, 0, 0);
                rr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aWh[s], pl, rr, 0, 0, 0);
                rr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aWl2[s], ph, rr, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int g = (r & 3) + 8 * (r >> 2) + 4 * hh;
                *(bf16_t*)(tH + g * ROWB + (32 * ct + c) * 2) = f2bf(rr[r]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (row + row_step < rows) fetch(row + row_step);
        for (int idx = lane; idx < total; idx += 64) {
            const int h = idx / nchunk, ck = idx - h * nchunk;
            const int64_t dst = base + h * hs + ck * 8;
            if (p.P) *(u32x4*)(p.P + dst) = *(const u32x4*)(tP + h * ROWB + ck * 16);
            *(u32x4*)(p.R + dst) = *(const u32x4*)(tH + h * ROWB + ck * 16);
        }
    }
}

// backward of the stage above.  dP = Ww^T dR ; dA = P o (dP - sum_j P dP) ; dS = Wl^T dA
// dWw[g,h] += sum dR_g P_h ; dWl[g,h] += sum dA_g S_h     (32x32x16 MFMA over the key axis, per-wave accumulators)
struct SoftmaxBwd {
    const bf16_t* dR;        // [B,H,N,Np]
    const bf16_t* P;         // [B,H,N,Np] (post-softmax, pre conv_w)
    const float* S;          // [B,H,N,Np] raw scores (only read when Wl is set); bf16 when s_bf16
    int s_bf16;
    const float* Wl;
    const float* Ww;
    bf16_t* dS;              // [B,H,N,Np]
    float* dWl;              // [H,H] += (may be null)
    float* dWw;
    int B, N, Np;
    unsigned long long* stamps;   // profiling only (dclip_trace_attn_stamps): 8 x u64 per (wave, row iteration < 4), else null
    int H;                        // (run-time copy: the kernel without head mixing takes any head count)
};

// Head-mixing softmax backward, everything matrix-shaped on v_mfma_f32_32x32x16_bf16 (one wave per (b, query row)):
//   Cw[g,h] = sum_j dR[g,j] P[h,j]                (this row's dW_w contribution; also gives the softmax row sums:)
//   rs[h]   = sum_j P[h,j] dP[h,j] = sum_g Ww[g,h] Cw[g,h]
//   dP = Ww^T dR ; dA = P o (dP - rs) ; dS = Wl^T dA ; dWl += dA S^T
// Tiles [32 heads][COLS keys] bf16 live in wave-private LDS; dA feeds the second mix straight from the accumulator
// registers (guide §3 "An accumulator tile as the next MFMA's operand": Wl^T is pre-permuted in k).
template <int H, int NS>
__global__ __launch_bounds__(256) void attn_softmax_bwd_mix_kernel(SoftmaxBwd p) {
    constexpr int COLS = 64 * NS, ROWB = COLS * 2 + 16, NCT = COLS / 32, TILE = 32 * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* tR = smem + wave * 4 * TILE;      // dR
    char* tP = tR + TILE;                   // P
    char* tS = tP + TILE;                   // S (bf16)
    char* tD = tS + TILE;                   // dA
    for (int idx = lane; idx < 4 * TILE / 16; idx += 64) ((u32x4*)tR)[idx] = u32x4{0u, 0u, 0u, 0u};
    const int hh = lane >> 5, c = lane & 31;
    bf16x8 aWw[2], aWl[2], aI[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = 16 * s + 8 * hh + e;                               // natural k order
            const int rho = 16 * s + 8 * (e >> 2) + 4 * hh + (e & 3);        // k order of an accumulator-as-operand
            aWw[s][e] = f2bf((g < H && c < H) ? p.Ww[g * H + c] : 0.f);      // A[h = c][k = g]   = Ww[g][h]
            aWl[s][e] = f2bf((rho < H && c < H) ? p.Wl[rho * H + c] : 0.f);  // A[h' = c][k = g]  = Wl[g][h']
            aI[s][e] = f2bf(g == c ? 1.f : 0.f);
        }
    float wwc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int g = (r & 3) + 8 * (r >> 2) + 4 * hh;
        wwc[r] = (g < H && c < H) ? p.Ww[g * H + c] : 0.f;
    }
    f32x16 accw = {0}, accl = {0};
    const int64_t hs = (int64_t)p.N * p.Np;
    const int rows = p.B * p.N;
    const int nchunk = p.Np >> 3, total = H * nchunk;
    const int nct = (p.N + 31) >> 5;                 // key tiles that hold real keys (the rest is all padding)
    const int nks = nct * 2;
    // NS == 2 (one workgroup per CU, one wave per SIMD): the next row's operands are fetched into registers while this row is
    // being computed, otherwise every row pays the full HBM latency before its first MFMA (229 -> 200 us at H = 12, N = 77).
    // With two workgroups per CU (NS == 1) the second wave already covers that latency and the extra registers cost more.
    constexpr bool PREFETCH = NS == 2;
    constexpr int NIT = (H * (COLS / 8) + 63) / 64;
    u32x4 qR[NIT], qP[NIT];
    float4 qS0[NIT], qS1[NIT];
    auto fetch = [&](int row) {
        const int64_t base = ((int64_t)(row / p.N) * H * p.N + row % p.N) * p.Np;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = lane + it * 64;
            if (idx < total) {
                const int h = idx / nchunk, ck = idx - h * nchunk;
                const int64_t src = base + h * hs + ck * 8;
                qR[it] = *(const u32x4*)(p.dR + src);
                qP[it] = *(const u32x4*)(p.P + src);
                if (p.s_bf16) qS0[it] = *(const float4*)((const bf16_t*)p.S + src);       // 8 bf16 scores in one 16-byte register set
                else { qS0[it] = *(const float4*)(p.S + src); qS1[it] = *(const float4*)(p.S + src + 4); }
            }
        }
    };
    const int row_first = blockIdx.x * 4 + wave, row_step = gridDim.x * 4;
    if (PREFETCH && row_first < rows) fetch(row_first);
    int iter = 0;
    auto stamp = [&](int k) {
        if (p.stamps && lane == 0 && iter < 4) {
            if (k == 1 || k == 4) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            p.stamps[(((int64_t)blockIdx.x * 4 + wave) * 4 + iter) * 8 + k] = __builtin_readcyclecounter();
        }
    };
    // dS of a row is staged in the dR tile (each key tile's columns are dead once its dP product has been read) and leaves as
    // 16-byte row segments at the START of the next iteration, after that row's loads have been issued: vmcnt retires loads and
    // stores in one in-order queue, so the 32 scattered 2-byte stores per row of the first version, issued before the next
    // row's loads, made every row wait for their acknowledgement.
    int64_t prev_base = -1;
    auto flush = [&]() {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = lane + it * 64;
            if (idx < total) {
                const int h = idx / nchunk, ck = idx - h * nchunk;
                *(u32x4*)(p.dS + prev_base + h * hs + ck * 8) = *(const u32x4*)(tR + h * ROWB + ck * 16);
            }
        }
    };
    for (int row = row_first; row < rows; row += row_step, ++iter) {
        const int b = row / p.N, i = row % p.N;
        const int64_t base = ((int64_t)b * H * p.N + i) * p.Np;
        __builtin_amdgcn_wave_barrier();
        stamp(0);
        if (!PREFETCH) fetch(row);                       // loads first ...
        if (prev_base >= 0) flush();                     // ... then the previous row's stores
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = lane + it * 64;
                if (idx < total) {
                    const int h = idx / nchunk, ck = idx - h * nchunk;
                    *(u32x4*)(tR + h * ROWB + ck * 16) = qR[it];
                    *(u32x4*)(tP + h * ROWB + ck * 16) = qP[it];
                    const float4 s0 = qS0[it], s1 = qS1[it];
                    if (p.s_bf16) *(float4*)(tS + h * ROWB + ck * 16) = s0;
                    else *(bf16x8*)(tS + h * ROWB + ck * 16) = bf16x8{f2bf(s0.x), f2bf(s0.y), f2bf(s0.z), f2bf(s0.w),
                                                                      f2bf(s1.x), f2bf(s1.y), f2bf(s1.z), f2bf(s1.w)};
                }
            }
        }
        prev_base = base;
        if (PREFETCH && row + row_step < rows) fetch(row + row_step);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        stamp(1);
        // Cw = dR P^T  (this row's dW_w contribution) and the softmax row sums rs[h] = sum_g Ww[g,h] Cw[g,h]
        float part = 0.f;
        {
            f32x16 cw = {0};
#pragma unroll
            for (int ks = 0; ks < COLS / 16; ++ks)
                if (ks < nks) {
                    const int off = c * ROWB + (ks * 16 + hh * 8) * 2;
                    cw = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(tR + off), *(const bf16x8*)(tP + off), cw, 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) part = fmaf(wwc[r], cw[r], part);
            accw += cw;
        }
        part += __shfl_xor(part, 32);                      // lanes h and h + 32 now hold rs[h]
        float rsr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int h0 = (r & 3) + 8 * (r >> 2);
            rsr[r] = hh ? lane_bcast(part, h0 + 4) : lane_bcast(part, h0);
        }
        stamp(2);
        // one key tile at a time: dP = Ww^T dR, P in accumulator layout, dA = P o (dP - rs), dS = Wl^T dA
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            if (ct < nct) {
                f32x16 dp = {0}, pa = {0};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aWw[s], tr_frag32<ROWB>(tR, 16 * s, 32 * ct, lane), dp, 0, 0, 0);
                    pa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aI[s], tr_frag32<ROWB>(tP, 16 * s, 32 * ct, lane), pa, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int h = (r & 3) + 8 * (r >> 2) + 4 * hh;
                    dp[r] = pa[r] * (dp[r] - rsr[r]);                                   // dA
                    *(bf16_t*)(tD + h * ROWB + (32 * ct + c) * 2) = f2bf(dp[r]);
                }
                f32x16 ds = {0};
                bf16x8 bf[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) bf[s][e] = f2bf(dp[8 * s + e]);
                    ds = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aWl[s], bf[s], ds, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int h = (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (h < H) *(bf16_t*)(tR + h * ROWB + (32 * ct + c) * 2) = f2bf(ds[r]);      // this key tile's dR columns are dead
                }
                // operands built by the VALU stay live past their (queued) MFMAs: attention_mix.hip, hw::keep_alive
                asm volatile("" :: "v"(bf[0]), "v"(bf[1]));
            }
        }
        // dW_l += dA S^T
        stamp(3);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ks = 0; ks < COLS / 16; ++ks)
            if (ks < nks) {
                const int off = c * ROWB + (ks * 16 + hh * 8) * 2;
                accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(tD + off), *(const bf16x8*)(tS + off), accl, 0, 0, 0);
            }
        stamp(4);
    }
    if (prev_base >= 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        flush();
    }
    // accumulators: element (g = (r&3) + 8*(r>>2) + 4*hh, h = c).  The 2 H^2 gradient elements live in ~36 cache lines that every
    // wave of the grid adds to: the workgroup's four waves are summed through LDS first (the tiles are dead) so that one wave
    // issues the atomics — same-line atomics serialise, and with 2 048 waves adding they were ~half of the kernel's time.
    __syncthreads();
    float* red = (float*)smem;                          // [4 waves][2][16][64]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        red[((wave * 2 + 0) * 16 + r) * 64 + lane] = accw[r];
        red[((wave * 2 + 1) * 16 + r) * 64 + lane] = accl[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int g = (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (g < H && c < H) {
                float w = 0.f, l = 0.f;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    w += red[((v * 2 + 0) * 16 + r) * 64 + lane];
                    l += red[((v * 2 + 1) * 16 + r) * 64 + lane];
                }
                if (p.dWw) unsafeAtomicAdd(p.dWw + g * H + c, w);
                if (p.dWl) unsafeAtomicAdd(p.dWl + g * H + c, l);
            }
        }
    }
}

// plain multi-head softmax backward (no head mixing): dS = P o (dR - sum_j P dR), one wave per (b, query row)
template <int NS>
__global__ __launch_bounds__(256) void attn_softmax_bwd_plain_kernel(SoftmaxBwd p) {
    const int H = p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t hs = (int64_t)p.N * p.Np;
    const int rows = p.B * p.N;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const int b = row / p.N, i = row % p.N;
        const int64_t base = ((int64_t)b * H * p.N + i) * p.Np;
        for (int h = 0; h < H; ++h) {
            float dr[NS], pv[NS], rs = 0.f;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int j = lane + 64 * s;
                dr[s] = j < p.N ? bf2f(p.dR[base + h * hs + j]) : 0.f;
                pv[s] = j < p.N ? bf2f(p.P[base + h * hs + j]) : 0.f;
                rs += dr[s] * pv[s];
            }
            rs = wave_sum(rs);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int j = lane + 64 * s;
                if (j < p.Np) p.dS[base + h * hs + j] = f2bf(pv[s] * (dr[s] - rs));
            }
        }
    }
}

unsigned long long* g_attn_stamps = nullptr;

int check_mm(const AttnMM& p, const char* who) {
    DCLIP_REQUIRE(p.A && p.Bm && p.C, "%s: null operand", who);
    DCLIP_REQUIRE(p.B > 0 && p.H > 0 && p.N > 0 && p.N <= NMAX, "%s: need 0 < N <= %d (N=%d)", who, NMAX, p.N);
    DCLIP_REQUIRE(p.hd == 32 || p.hd == 64, "%s: head dim must be 32 or 64 (got %d)", who, p.hd);
    DCLIP_REQUIRE(p.Np % 8 == 0 && p.Np >= p.N, "%s: Np must be a multiple of 8 and >= N", who);
    return DCLIP_OK;
}

}  // namespace

extern "C" int dclip_trace_attn_stamps(void* buf) { g_attn_stamps = (unsigned long long*)buf; return 0; }

extern "C" int dclip_attn_nt(const void* A, int64_t lda, const void* Bm, int64_t ldb, void* C, int out_f32, int64_t B,
                             int64_t H, int64_t N, int64_t Np, int64_t hd, float alpha, void* stream) {
    AttnMM p{A, lda, (const bf16_t*)Bm, ldb, C, Np, (int)B, (int)H, (int)N, (int)Np, (int)hd, alpha, 0};
    if (int rc = check_mm(p, "dclip_attn_nt")) return rc;
    DCLIP_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, "dclip_attn_nt: token-major strides must be multiples of 8");
    TraceScope tr(DCLIP_TRACE_ATTN, 2.0 * B * H * N * N * hd, 4.0 * B * H * N * hd + (out_f32 ? 4.0 : 2.0) * B * H * N * Np, stream, (int)(B * H), (int)N, (int)hd, 1);
    const dim3 grid((unsigned)((B * H + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    if (hd == 32) {
        if (out_f32) hipLaunchKernelGGL((attn_nt_kernel<32, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((attn_nt_kernel<32, false>), grid, dim3(256), 0, st, p);
    } else {
        if (out_f32) hipLaunchKernelGGL((attn_nt_kernel<64, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((attn_nt_kernel<64, false>), grid, dim3(256), 0, st, p);
    }
    return dclip_check_launch("dclip_attn_nt");
}

extern "C" int dclip_attn_nn(const void* A, const void* Bm, int64_t ldb, void* C, int64_t ldc, int64_t B, int64_t H,
                             int64_t N, int64_t Np, int64_t hd, float alpha, int a_blocked, void* stream) {
    AttnMM p{A, Np, (const bf16_t*)Bm, ldb, C, ldc, (int)B, (int)H, (int)N, (int)Np, (int)hd, alpha, a_blocked};
    if (int rc = check_mm(p, "dclip_attn_nn")) return rc;
    DCLIP_REQUIRE(ldc % 8 == 0 && ((uintptr_t)C % 16) == 0 && ldb % 8 == 0 && ((uintptr_t)Bm % 16) == 0, "dclip_attn_nn: token-major operands must be 16-byte aligned");
    TraceScope tr(DCLIP_TRACE_ATTN, 2.0 * B * H * N * N * hd, 4.0 * B * H * N * hd + 2.0 * B * H * N * Np, stream, (int)(B * H), (int)N, (int)hd, 2);
    const dim3 grid((unsigned)((B * H + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    const size_t n32 = ((size_t)N + 31) & ~(size_t)31;
    if (hd == 32) hipLaunchKernelGGL((attn_nn_kernel<32>), grid, dim3(256), 4 * n32 * (32 * 2 + 32), st, p);
    else hipLaunchKernelGGL((attn_nn_kernel<64>), grid, dim3(256), 4 * n32 * (64 * 2 + 32), st, p);
    return dclip_check_launch("dclip_attn_nn");
}

extern "C" int dclip_attn_tn(const void* A, const void* Bm, int64_t ldb, void* C, int64_t ldc, int64_t B, int64_t H,
                             int64_t N, int64_t Np, int64_t hd, float alpha, int a_blocked, void* stream) {
    AttnMM p{A, Np, (const bf16_t*)Bm, ldb, C, ldc, (int)B, (int)H, (int)N, (int)Np, (int)hd, alpha, a_blocked};
    if (int rc = check_mm(p, "dclip_attn_tn")) return rc;
    DCLIP_REQUIRE(ldc % 8 == 0 && ((uintptr_t)C % 16) == 0 && ldb % 8 == 0 && ((uintptr_t)Bm % 16) == 0, "dclip_attn_tn: token-major operands must be 16-byte aligned");
    TraceScope tr(DCLIP_TRACE_ATTN, 2.0 * B * H * N * N * hd, 4.0 * B * H * N * hd + 2.0 * B * H * N * Np, stream, (int)(B * H), (int)N, (int)hd, 3);
    const dim3 grid((unsigned)((B * H + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    const int ntile = ((int)Np + 15) / 16;
    const int mj = ntile <= 4 ? 4 : (ntile <= 5 ? 5 : 8);
    const size_t lds = (size_t)4 * (32 * (mj * 32 + 32) + 32 * ((int)hd * 2 + 32));
#define TN_LAUNCH(HDv, NTv) hipLaunchKernelGGL((attn_tn_kernel<HDv, NTv>), grid, dim3(256), lds, st, p)
    if (hd == 32) { if (ntile <= 4) TN_LAUNCH(32, 4); else if (ntile <= 5) TN_LAUNCH(32, 5); else TN_LAUNCH(32, 8); }
    else { if (ntile <= 4) TN_LAUNCH(64, 4); else if (ntile <= 5) TN_LAUNCH(64, 5); else TN_LAUNCH(64, 8); }
#undef TN_LAUNCH
    return dclip_check_launch("dclip_attn_tn");
}

extern "C" int dclip_attn_fused_fwd(const void* qkv, int64_t ldq, void* ctx, int64_t ldc, int64_t B, int64_t H, int64_t N, int64_t hd,
                                    float scale, int causal, void* stream) {
    DCLIP_REQUIRE(qkv && ctx && B > 0 && H > 0 && N > 0 && N <= NMAX, "dclip_attn_fused_fwd: bad argument (N <= %d)", NMAX);
    DCLIP_REQUIRE(hd == 32 || hd == 64, "dclip_attn_fused_fwd: head dim must be 32 or 64 (got %ld)", (long)hd);
    DCLIP_REQUIRE(ldq % 8 == 0 && ldc % 8 == 0 && ((uintptr_t)qkv % 16) == 0 && ((uintptr_t)ctx % 16) == 0, "dclip_attn_fused_fwd: misaligned buffers");
    AttnFused p{(const bf16_t*)qkv, ldq, (bf16_t*)ctx, ldc, (int)B, (int)H, (int)N, causal, scale, 1};
    TraceScope tr(DCLIP_TRACE_ATTN, 4.0 * B * H * N * N * hd, 8.0 * B * H * N * hd, stream, (int)(B * H), (int)N, (int)hd, 4);
    hipStream_t st = (hipStream_t)stream;
    const int ntile = ((int)N + 15) / 16;
    // an instance holds registers for NT key tiles; odd tile counts need an instance with the 16-key tail step (NT odd)
    const int NT = ntile <= 4 ? 4 : (ntile == 5 ? 5 : (ntile == 7 ? 7 : 8));
    const int nrows = (NT & 1) ? ntile * 16 : ((ntile + 1) & ~1) * 16;            // (even instances: whole 32-key steps, zero rows behind N)
    const size_t vtile = (size_t)nrows * (hd * 2 + 32), ptile = 16 * ((size_t)nrows * 2 + 16);
    // workgroup shape: `ppw` problems of `split` waves each (split = 2: two waves share a problem's V tile and take alternate query
    // tiles), whichever puts the most waves on a CU (160 KB of LDS; registers allow 12 / 8 waves: launch bounds); ties go to the split
    // form, which also halves the dependent chain of a problem
    const int reg_cap = NT <= 7 ? 12 : 8;
    int nw = 4, split = 1, best = -1;
    for (int sp = 2; sp >= 1; --sp) {
        if (sp == 2 && ntile < 2) continue;
        for (int ppw = 4 / sp; ppw >= 1; ppw >>= 1) {
            const size_t wg = ppw * vtile + (size_t)ppw * sp * ptile;
            int on_cu = (int)((160 * 1024) / wg) * ppw * sp;
            on_cu = on_cu > reg_cap ? reg_cap : on_cu;
            if (on_cu > best) { best = on_cu; nw = ppw * sp; split = sp; }
        }
    }
    p.split = split;
    const int ppw = nw / split;
    const dim3 grid((unsigned)((B * H + ppw - 1) / ppw));
    const size_t lds = ppw * vtile + (size_t)nw * ptile;
#define FUSED_LAUNCH(HDv, NTv) hipLaunchKernelGGL((attn_fused_fwd_kernel<HDv, NTv>), grid, dim3(64 * nw), lds, st, p)
    if (hd == 32) { if (NT == 4) FUSED_LAUNCH(32, 4); else if (NT == 5) FUSED_LAUNCH(32, 5); else if (NT == 7) FUSED_LAUNCH(32, 7); else FUSED_LAUNCH(32, 8); }
    else { if (NT == 4) FUSED_LAUNCH(64, 4); else if (NT == 5) FUSED_LAUNCH(64, 5); else if (NT == 7) FUSED_LAUNCH(64, 7); else FUSED_LAUNCH(64, 8); }
#undef FUSED_LAUNCH
    return dclip_check_launch("dclip_attn_fused_fwd");
}

#define SM_DISPATCH_H(Hv, NSv, ...)                                               \
    switch (Hv) {                                                                 \
        case 2: { constexpr int HH = 2; SM_DISPATCH_NS(NSv, __VA_ARGS__); break; }   \
        case 4: { constexpr int HH = 4; SM_DISPATCH_NS(NSv, __VA_ARGS__); break; }   \
        case 8: { constexpr int HH = 8; SM_DISPATCH_NS(NSv, __VA_ARGS__); break; }   \
        case 12: { constexpr int HH = 12; SM_DISPATCH_NS(NSv, __VA_ARGS__); break; } \
        case 24: { constexpr int HH = 24; SM_DISPATCH_NS(NSv, __VA_ARGS__); break; } \
        default: dclip_set_error("attention softmax: unsupported head count %d (2/4/8/12/24)", (int)(Hv)); return DCLIP_EINVAL; \
    }
#define SM_DISPATCH_NS(NSv, ...)                              \
    if ((NSv) == 1) { constexpr int NSS = 1; __VA_ARGS__; }   \
    else { constexpr int NSS = 2; __VA_ARGS__; }

extern "C" int dclip_attn_softmax_fwd(const float* S, const float* Wl, const float* Ww, void* P, void* R, int64_t B, int64_t H,
                                      int64_t N, int64_t Np, int causal, void* stream) {
    DCLIP_REQUIRE(S && R && B > 0 && N > 0 && N <= NMAX && Np % 8 == 0 && Np >= N, "dclip_attn_softmax_fwd: bad argument");
    DCLIP_REQUIRE((Wl == nullptr) == (Ww == nullptr), "dclip_attn_softmax_fwd: conv_l and conv_w come together");
    SoftmaxFwd p{S, Wl, Ww, (bf16_t*)P, (bf16_t*)R, (int)B, (int)N, (int)Np, causal, (int)H};
    TraceScope tr(DCLIP_TRACE_ATTN, Wl ? 4.0 * B * H * H * N * N : 0.0, (4.0 + 2.0 + (P ? 2.0 : 0.0)) * B * H * N * Np, stream, (int)(B * H), (int)N, (int)H, 5);
    const int ns = N > 64 ? 2 : 1;
    hipStream_t st = (hipStream_t)stream;
    if (Wl && !causal && H > 12) {     // H <= 12: the 144-FMA register mix is faster than 32-row MFMA tiles (measured)
        int blocks = (int)((B * N + 3) / 4);
        if (blocks > 512) blocks = 512;              // persistent waves: constant fragments / LDS zero-fill amortised over rows
        const size_t lds = (size_t)4 * 3 * 32 * (64 * ns * 2 + 16);
        SM_DISPATCH_H(H, ns, hipLaunchKernelGGL((attn_softmax_fwd_mix_kernel<HH, NSS>), dim3(blocks), dim3(256), lds, st, p));
    } else if (!Wl) {                  // no head mixing: any head count
        const dim3 grid((unsigned)((B * N + 3) / 4));
        if (ns == 1) hipLaunchKernelGGL((attn_softmax_fwd_plain_kernel<1>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((attn_softmax_fwd_plain_kernel<2>), grid, dim3(256), 0, st, p);
    } else {
        const dim3 grid((unsigned)((B * N + 3) / 4));
        SM_DISPATCH_H(H, ns, hipLaunchKernelGGL((attn_softmax_fwd_kernel<HH, NSS>), grid, dim3(256), 0, st, p));
    }
    return dclip_check_launch("dclip_attn_softmax_fwd");
}

extern "C" int dclip_attn_softmax_bwd(const void* dR, const void* P, const void* S, int scores_bf16, const float* Wl, const float* Ww,
                                      void* dS, float* dWl, float* dWw, int64_t B, int64_t H, int64_t N, int64_t Np, void* stream) {
    DCLIP_REQUIRE(dR && P && dS && B > 0 && N > 0 && N <= NMAX && Np % 8 == 0 && Np >= N, "dclip_attn_softmax_bwd: bad argument");
    DCLIP_REQUIRE((Wl == nullptr) == (Ww == nullptr), "dclip_attn_softmax_bwd: conv_l and conv_w come together");
    DCLIP_REQUIRE(!Wl || S, "dclip_attn_softmax_bwd: raw scores needed for dW_l");
    SoftmaxBwd p{(const bf16_t*)dR, (const bf16_t*)P, (const float*)S, scores_bf16, Wl, Ww, (bf16_t*)dS, dWl, dWw, (int)B, (int)N, (int)Np, g_attn_stamps, (int)H};
    TraceScope tr(DCLIP_TRACE_ATTN, Wl ? 8.0 * B * H * H * N * N : 0.0, (2.0 + 2.0 + 2.0 + (Wl ? (scores_bf16 ? 2.0 : 4.0) : 0.0)) * B * H * N * Np, stream, (int)(B * H), (int)N, (int)H, 6);
    int blocks = (int)((B * N + 3) / 4);
    if (blocks > 2048) blocks = 2048;
    const int ns = N > 64 ? 2 : 1;
    hipStream_t st = (hipStream_t)stream;
    if (Wl) {
        // ~280 registers -> one resident workgroup per CU: launch one persistent workgroup per CU so the per-workgroup
        // setup (LDS zero-fill, constant Ww / Wl fragments) is amortised over all its rows
        const size_t lds = (size_t)4 * 4 * 32 * (64 * ns * 2 + 16);
        const int per_cu = lds <= 80 * 1024 ? 2 : 1;        // persistent workgroups: setup amortised over all rows
        if (blocks > 256 * per_cu) blocks = 256 * per_cu;
        SM_DISPATCH_H(H, ns, hipLaunchKernelGGL((attn_softmax_bwd_mix_kernel<HH, NSS>), dim3(blocks), dim3(256), lds, st, p));
    } else {
        if (ns == 1) hipLaunchKernelGGL((attn_softmax_bwd_plain_kernel<1>), dim3(blocks), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((attn_softmax_bwd_plain_kernel<2>), dim3(blocks), dim3(256), 0, st, p);
    }
    return dclip_check_launch("dclip_attn_softmax_bwd");
}
