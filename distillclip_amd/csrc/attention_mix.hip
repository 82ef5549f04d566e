// Head-mixed student attention with the score tensors kept in registers.
//
//   reference: model/component/weight_share_model.py:101-125 (MiniAttention.forward)
//       S_h = scale * Q_h K_h^T ; A = conv_l(S) (1x1 conv over the head channel) ; P = softmax_j(A) ; R = conv_w(P) ; ctx_h = R_h V_h
//
// The unfused path (attention.hip) moves S (f32), P, R, dR, dS through HBM: 14 + 18 bytes per score element and step.  Here
// one WAVE owns (sample b, 16 query rows) and walks the keys in blocks of 16.  S^T = K Q^T comes out of the MFMA in the
// accumulator layout (lane <-> query i0 + (lane & 15), registers <-> keys 16 jb + 4 (lane >> 4) + r), and in that layout a lane
// holds ALL heads of its (query, key) elements: both head mixes are plain register FMAs against wave-uniform weights (scalar
// operands), the softmax statistics of a query are a reduction over the lane's registers, the key blocks and the 4 lanes that
// share lane & 15.  S, A, P and dR never exist in memory:
//
//   forward   pass 1: S, A per key block -> running max / sum per (head, query)      pass 2: S, A, P, R -> R (bf16) + statistics
//   backward  pass A: S, A, P, dR = V dO^T, dP = conv_w^T(dR) -> sum_j P dP           pass B: ... dA, dS = conv_l^T(dA) -> dS (bf16),
//             dW_w += dR P^T and dW_l += dA S^T on the matrix pipe (operands transposed through a small wave-private LDS tile)
//
// R is written once (the forward's R V product and the backward's dV = R^T dO read it) and dS once (dQ, dK): 4 + 6 bytes per
// element instead of 32.  The products that contract over keys or queries (R V, R^T dO, dS K, dS^T Q) stay in attention.hip.
#include <math.h>
#include <stdlib.h>
#include "common.h"

namespace {

constexpr float NEG_BIG = -1e30f;

struct MixFwd {
    const bf16_t* qkv; int64_t ld;       // [B*N, 3D]: q | k | v, head h at column h * HD
    const float* Wl; const float* Ww;    // [H, H] conv_l / conv_w weights (f32 masters)
    bf16_t* R;                           // [B, H, N, Np] mixed probabilities, pad columns zero
    float* stats;                        // [B, H, N]: log-sum-exp of row (b, h, i) of A  (P = exp(A - lse))
    int B, N, Np, D, QT;
    float scale;
};

struct MixBwd {
    const bf16_t* qkv; int64_t ld;
    const bf16_t* dO; int64_t ldo;       // [B*N, D] gradient of ctx
    const float* Wl; const float* Ww;
    const float* stats;                  // [B, H, N] log-sum-exp rows of the forward
    bf16_t* dS;                          // [B, H, N, Np] gradient of the (scaled) pre-mix scores, pad columns zero
    float* dWl; float* dWw;              // [H, H] += (f32 atomics)
    int B, N, Np, D, QT;
    float scale;
};

// out[g] = sum_h W[g, h] in[h]      (W wave-uniform: scalar loads, one SGPR operand per FMA)
template <int H>
__device__ __forceinline__ void mix_rows(const f32x4 (&in)[H], f32x4 (&out)[H], const float* __restrict__ W) {
#pragma unroll
    for (int g = 0; g < H; ++g) {
        f32x4 a = in[0] * W[g * H];
#pragma unroll
        for (int h = 1; h < H; ++h) a += in[h] * W[g * H + h];
        out[g] = a;
    }
}
// out[h] = sum_g W[g, h] in[g]      (the adjoint)
template <int H>
__device__ __forceinline__ void mix_cols(const f32x4 (&in)[H], f32x4 (&out)[H], const float* __restrict__ W) {
#pragma unroll
    for (int h = 0; h < H; ++h) {
        f32x4 a = in[0] * W[h];
#pragma unroll
        for (int g = 1; g < H; ++g) a += in[g] * W[g * H + h];
        out[h] = a;
    }
}

// S^T block: s[h][r] = scale * q_{i0 + c, h} . k_{16 jb + 4 g4 + r, h}      (A operand = K rows, B operand = Q rows)
template <int H, int HD>
__device__ __forceinline__ void scores_block(const bf16_t* __restrict__ krow, const bf16_t* __restrict__ qrow, int g4, float scale,
                                             f32x4 (&s)[H]) {
    constexpr int KS = HD / 32;
#pragma unroll
    for (int h = 0; h < H; ++h) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 kf = *(const bf16x8*)(krow + h * HD + ks * 32 + g4 * 8);
            const bf16x8 qf = *(const bf16x8*)(qrow + h * HD + ks * 32 + g4 * 8);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, acc, 0, 0, 0);
        }
        s[h] = acc * scale;
    }
}

// out[g] = sum_h W[g, h] in[h]  (ROWS)   /   out[h] = sum_g W[g, h] in[g]  (COLS = the adjoint), [H][2] tensors, W in LDS.
// The weights of output row o + 1 are read (broadcast ds_reads) while row o is accumulated, and a scheduling barrier per row keeps
// the compiler from hoisting all H^2 weights into registers (spills) or serialising load -> use per row (LDS latency exposed).
template <int H, bool COLS>
__device__ __forceinline__ void mix2(const float (&in)[H][2], float (&out)[H][2], const float* W) {
    float wc[H], wn[H];
#pragma unroll
    for (int k = 0; k < H; ++k) wc[k] = COLS ? W[k * H] : W[k];
#pragma unroll
    for (int o = 0; o < H; ++o) {
        if (o + 1 < H) {
#pragma unroll
            for (int k = 0; k < H; ++k) wn[k] = COLS ? W[k * H + o + 1] : W[(o + 1) * H + k];
        }
        float a0 = in[0][0] * wc[0], a1 = in[0][1] * wc[0];
#pragma unroll
        for (int k = 1; k < H; ++k) { a0 += in[k][0] * wc[k]; a1 += in[k][1] * wc[k]; }
        out[o][0] = a0; out[o][1] = a1;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < H; ++k) wc[k] = wn[k];
    }
}

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// One key block of an operand ([16 keys][D] bf16 rows of the packed qkv matrix, column offset col0) -> LDS by LDS-DMA, no padding:
// 16-byte chunk cc of row r lands at chunk position r * (D / 8) + (cc ^ r), so that the 16 lanes of a fragment read (16 rows,
// same chunk) hit 16 different 16-byte slots of the 256-byte bank row (D * 2 is a multiple of 256).  LDS-DMA writes lane-linear:
// the permutation is applied to the SOURCE address.  Wave w of nw issues the wave-instructions w, w + nw, ...
__device__ __forceinline__ void stage_keys(const bf16_t* __restrict__ base, int64_t ld, int col0, int D, int N, int jb, char* buf,
                                           int wave, int nw, int lane) {
    const int cpr = D >> 3;                          // 16-byte chunks per row
    const int ninst = (16 * cpr) >> 6;               // wave-instructions per tile (D % 64 == 0)
    for (int k = wave; k < ninst; k += nw) {
        const int pos = k * 64 + lane;
        const int r = pos / cpr, cp = pos - r * cpr;
        const int cc = cp ^ r;                       // r < 16 and cpr % 16 == 0: stays inside the row
        const int row = min(jb * 16 + r, N - 1);
        const bf16_t* src = base + (int64_t)row * ld + col0 + cc * 8;
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(buf + k * 1024), 16, 0, 0);
    }
}
// fragment (A operand: lane & 15 = key row, 8 contraction elements from column `col`) of a tile staged by stage_keys
__device__ __forceinline__ bf16x8 key_frag(const char* buf, int D, int c, int col) {
    const int cpr = D >> 3;
    return *(const bf16x8*)(buf + ((c * cpr + ((col >> 3) ^ c)) << 4));
}

// Forward: one WORKGROUP per sample, one wave per 16-query tile.  The key block K[jb] ([16][D], all heads) is shared by the
// waves through a double-buffered LDS tile filled by LDS-DMA one block ahead; each wave keeps the B-operand fragments of its 16
// queries (all heads) in registers for the whole sample.
template <int H, int HD>
__global__ __launch_bounds__(512) void attn_mix_fwd_kernel(MixFwd p, const float* __restrict__ Wl_g, const float* __restrict__ Ww_g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = HD / 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int b = blockIdx.x, it = wave;
    const int c = lane & 15, g4 = lane >> 4;
    const int i = it * 16 + c;
    const bool iok = i < p.N;
    const bf16_t* base = p.qkv + (int64_t)b * p.N * p.ld;
    const bf16_t* qrow = base + (int64_t)min(i, p.N - 1) * p.ld;
    const int nb = (p.N + 15) >> 4;
    const int tile_bytes = 16 * p.D * 2;

    // The mix weights are scalar operands (wave-uniform s_loads).  The 2 H^2 values do not fit the SGPR file, so every mix
    // re-fetches them through the scalar cache, and at 2 waves per SIMD that latency is what bounds this kernel (several times the
    // VALU time).  Through LDS broadcast reads instead, the compiler hoists them into VGPRs and spills (352 us against 247 us,
    // text student, B = 512); on the matrix pipe (4 x 4 lane-group transposes by v_permlane{16,32}_swap + 16x16x16 MFMA with the
    // mix matrix as a constant A operand) the weight fetch disappears: 171 us in a first version, not finished this round
    // (DESIGN.md section 7c).
    const float* __restrict__ Wl = Wl_g;
    const float* __restrict__ Ww = Ww_g;
    bf16x8 qf[H][KS];
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[h][ks] = *(const bf16x8*)(qrow + h * HD + ks * 32 + g4 * 8);
    float mp[H], sp[H];
#pragma unroll
    for (int g = 0; g < H; ++g) { mp[g] = NEG_BIG; sp[g] = 0.f; }

    stage_keys(base, p.ld, p.D, p.D, p.N, 0, smem, wave, nw, lane);
    __syncthreads();
    const int total = 2 * nb;                        // pass 1 (statistics) then pass 2 (probabilities, second mix, R)
    for (int t = 0; t < total; ++t) {
        const int jb = t < nb ? t : t - nb;
        const char* kt = smem + (t & 1) * tile_bytes;
        if (t + 1 < total) stage_keys(base, p.ld, p.D, p.D, p.N, (t + 1 < nb ? t + 1 : t + 1 - nb), smem + ((t + 1) & 1) * tile_bytes, wave, nw, lane);
        // two sub-steps of 2 of the lane's 4 keys each: [H][2] tensors keep the register budget at 2 waves per SIMD next to the
        // query fragments; the MFMAs of the block are simply issued twice
        bf16x2 rlo[H];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            float sc[H][2], ac[H][2];
#pragma unroll
            for (int h = 0; h < H; ++h) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(key_frag(kt, p.D, c, h * HD + ks * 32 + g4 * 8), qf[h][ks], acc, 0, 0, 0);
                sc[h][0] = acc[2 * sub] * p.scale; sc[h][1] = acc[2 * sub + 1] * p.scale;
                if (h % 4 == 3) __builtin_amdgcn_sched_barrier(0);       // bounded number of LDS fragment reads in flight
            }
            mix2<H, false>(sc, ac, Wl);                                  // A = conv_l(S)
            const int j0 = jb * 16 + g4 * 4 + 2 * sub;
            const bool v0 = j0 < p.N, v1 = j0 + 1 < p.N;
            if (t < nb) {
                // running maximum / sum of every (head, query) over this lane's keys
#pragma unroll
                for (int g = 0; g < H; ++g) {
                    float bm = v0 ? ac[g][0] : NEG_BIG;
                    bm = v1 ? fmaxf(bm, ac[g][1]) : bm;
                    const float mn = fmaxf(mp[g], bm);
                    float acc = sp[g] * __expf(mp[g] - mn);
                    acc += v0 ? __expf(ac[g][0] - mn) : 0.f;
                    acc += v1 ? __expf(ac[g][1] - mn) : 0.f;
                    sp[g] = acc; mp[g] = mn;
                }
            } else {
#pragma unroll
                for (int g = 0; g < H; ++g) {
                    ac[g][0] = v0 ? __expf(ac[g][0] - mp[g]) : 0.f;
                    ac[g][1] = v1 ? __expf(ac[g][1] - mp[g]) : 0.f;
                }
                mix2<H, false>(ac, sc, Ww);                              // R = conv_w(P)
#pragma unroll
                for (int g = 0; g < H; ++g) {
                    const bf16x2 rv = {f2bf(sc[g][0]), f2bf(sc[g][1])};
                    if (sub == 0) rlo[g] = rv;
                    else if (iok && j0 - 2 < p.Np) {
                        const bf16x4 o = {rlo[g][0], rlo[g][1], rv[0], rv[1]};
                        *(bf16x4*)(p.R + (((int64_t)b * H + g) * p.N + i) * p.Np + (j0 - 2)) = o;
                    }
                }
            }
        }
        if (t == nb - 1) {
            // end of pass 1: the 4 lanes that share lane & 15 hold the 4 key sub-rows of the same query
#pragma unroll
            for (int g = 0; g < H; ++g) {
                float m = fmaxf(mp[g], __shfl_xor(mp[g], 16));
                m = fmaxf(m, __shfl_xor(m, 32));
                float e = sp[g] * __expf(mp[g] - m);
                e += __shfl_xor(e, 16);
                e += __shfl_xor(e, 32);
                mp[g] = m + __logf(e);                   // log-sum-exp: P = exp(A - lse)
                if (g4 == 0 && iok) p.stats[((int64_t)b * H + g) * p.N + i] = mp[g];
            }
        }
        __syncthreads();       // the next tile has landed (the fence drains the LDS-DMA) and every wave is done with this one
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward.  Four [H][elements] tensors are live at once (S, P, dR, dP -> dA): with 4 elements per lane that is 16 H registers,
// too many next to the statistics for H = 12, so a key block is processed in two sub-steps of 2 elements per lane (the MFMAs of
// a block are simply issued again for the second half: they are a negligible part of the work).
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int RS = 2;                            // elements per lane and sub-step
constexpr int WG_ROWB = 128 * 2 + 16;            // wave-private LDS tile of the weight-gradient products: [16 head rows][128 elements]
constexpr int WG_TILE = 16 * WG_ROWB;            //   bf16, rows padded against bank conflicts

// out[g][h] (16 x 16 tile, accumulator layout) += sum_elem X[g][elem] Y[h][elem] over the 128 elements of this sub-step
template <int H>
__device__ __forceinline__ void outer_acc(const float (&x)[H][RS], const float (&y)[H][RS], char* tx, char* ty, int lane, f32x4& acc) {
    const int c = lane & 15, g4 = lane >> 4;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const bf16x2 xv = {f2bf(x[h][0]), f2bf(x[h][1])};
        const bf16x2 yv = {f2bf(y[h][0]), f2bf(y[h][1])};
        *(bf16x2*)(tx + h * WG_ROWB + lane * 4) = xv;
        *(bf16x2*)(ty + h * WG_ROWB + lane * 4) = yv;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 af = *(const bf16x8*)(tx + c * WG_ROWB + (ks * 32 + g4 * 8) * 2);
        const bf16x8 bf = *(const bf16x8*)(ty + c * WG_ROWB + (ks * 32 + g4 * 8) * 2);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc, 0, 0, 0);
    }
}

template <int H, int HD>
__global__ __launch_bounds__(256) void attn_mix_bwd_kernel(MixBwd p, const float* __restrict__ Wl_g, const float* __restrict__ Ww_g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* tx = smem + wave * 2 * WG_TILE;
    char* ty = tx + WG_TILE;
    const float* __restrict__ Wl = Wl_g;                 // scalar operands (see the forward kernel)
    const float* __restrict__ Ww = Ww_g;
    // rows >= H of the tiles are never written: zero them once so that the ignored part of the MFMA tile stays finite
    for (int idx = lane; idx < 2 * WG_TILE / 16; idx += 64) ((u32x4*)tx)[idx] = u32x4{0u, 0u, 0u, 0u};
    const int c = lane & 15, g4 = lane >> 4;
    constexpr int KS = HD / 32;
    f32x4 accW = {0.f, 0.f, 0.f, 0.f}, accL = {0.f, 0.f, 0.f, 0.f};
    const int nitem = p.B * p.QT;
    const int nb = (p.N + 15) >> 4;
    for (int item = blockIdx.x * 4 + wave; item < nitem; item += gridDim.x * 4) {
        const int b = item / p.QT, it = item - b * p.QT;
        const int i = it * 16 + c;
        const bool iok = i < p.N;
        const bf16_t* base = p.qkv + (int64_t)b * p.N * p.ld;
        const bf16_t* qrow = base + (int64_t)min(i, p.N - 1) * p.ld;
        const bf16_t* dorow = p.dO + ((int64_t)b * p.N + min(i, p.N - 1)) * p.ldo;
        float lse[H], rd[H];
#pragma unroll
        for (int g = 0; g < H; ++g) {
            // queries beyond N: P = 0 (lse = +big), so they drop out of every sum below
            lse[g] = iok ? p.stats[((int64_t)b * H + g) * p.N + i] : 1e30f;
            rd[g] = 0.f;
        }
        float sc[H][RS], pr[H][RS], dr[H][RS], dp[H][RS];
        // S, P, dR, dP of sub-step `sub` (elements 2 sub, 2 sub + 1 of the lane's 4 keys) of key block jb
        auto substep = [&](int jb, int sub) {
            const int jr = min(jb * 16 + c, p.N - 1);
            const bf16_t* krow = base + (int64_t)jr * p.ld + p.D;
            const bf16_t* vrow = krow + p.D;
            const int j0 = jb * 16 + g4 * 4 + sub * RS;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int o = h * HD + ks * 32 + g4 * 8;
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(krow + o), *(const bf16x8*)(qrow + o), a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(vrow + o), *(const bf16x8*)(dorow + o), a2, 0, 0, 0);
                }
                if (sub == 0) { sc[h][0] = a1[0] * p.scale; sc[h][1] = a1[1] * p.scale; dr[h][0] = a2[0]; dr[h][1] = a2[1]; }
                else { sc[h][0] = a1[2] * p.scale; sc[h][1] = a1[3] * p.scale; dr[h][0] = a2[2]; dr[h][1] = a2[3]; }
                __builtin_amdgcn_sched_barrier(0);           // one head's fragment loads in flight at a time (register budget)
            }
#pragma unroll
            for (int g = 0; g < H; ++g) {                    // A = conv_l(S) ; P = exp(A - lse)
                float a0 = sc[0][0] * Wl[g * H], a1 = sc[0][1] * Wl[g * H];
#pragma unroll
                for (int h = 1; h < H; ++h) { a0 += sc[h][0] * Wl[g * H + h]; a1 += sc[h][1] * Wl[g * H + h]; }
                pr[g][0] = (j0 < p.N) ? __expf(a0 - lse[g]) : 0.f;
                pr[g][1] = (j0 + 1 < p.N) ? __expf(a1 - lse[g]) : 0.f;
            }
#pragma unroll
            for (int h = 0; h < H; ++h) {                    // dP_h = sum_g Ww[g, h] dR_g
                float a0 = dr[0][0] * Ww[h], a1 = dr[0][1] * Ww[h];
#pragma unroll
                for (int g = 1; g < H; ++g) { a0 += dr[g][0] * Ww[g * H + h]; a1 += dr[g][1] * Ww[g * H + h]; }
                dp[h][0] = a0; dp[h][1] = a1;
            }
        };
        // ---- pass A: sum_j P dP per (head, query) -----------------------------------------------------------------------------
        for (int jb = 0; jb < nb; ++jb)
            for (int sub = 0; sub < 2; ++sub) {
                substep(jb, sub);
#pragma unroll
                for (int h = 0; h < H; ++h) rd[h] += pr[h][0] * dp[h][0] + pr[h][1] * dp[h][1];
            }
#pragma unroll
        for (int h = 0; h < H; ++h) {
            rd[h] += __shfl_xor(rd[h], 16);
            rd[h] += __shfl_xor(rd[h], 32);
        }
        // ---- pass B: dA, dS, weight gradients ------------------------------------------------------------------------------------
        for (int jb = 0; jb < nb; ++jb)
            for (int sub = 0; sub < 2; ++sub) {
                substep(jb, sub);
                outer_acc<H>(dr, pr, tx, ty, lane, accW);                    // dW_w[g, h] += sum dR_g P_h
#pragma unroll
                for (int h = 0; h < H; ++h) {                                // dA (softmax backward)
                    dp[h][0] = pr[h][0] * (dp[h][0] - rd[h]);
                    dp[h][1] = pr[h][1] * (dp[h][1] - rd[h]);
                }
                outer_acc<H>(dp, sc, tx, ty, lane, accL);                    // dW_l[g, h] += sum dA_g S_h
                const int j0 = jb * 16 + g4 * 4 + sub * RS;
                if (iok && j0 < p.Np) {
#pragma unroll
                    for (int h = 0; h < H; ++h) {                            // dS_h = sum_g Wl[g, h] dA_g
                        float a0 = dp[0][0] * Wl[h], a1 = dp[0][1] * Wl[h];
#pragma unroll
                        for (int g = 1; g < H; ++g) { a0 += dp[g][0] * Wl[g * H + h]; a1 += dp[g][1] * Wl[g * H + h]; }
                        const bf16x2 o = {f2bf(a0), f2bf(a1)};
                        *(bf16x2*)(p.dS + (((int64_t)b * H + h) * p.N + i) * p.Np + j0) = o;
                    }
                }
            }
    }
    // accumulator layout: row g = 4 g4 + r, column h = c
    if (c < H) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int g = g4 * 4 + r;
            if (g < H) {
                unsafeAtomicAdd(p.dWw + g * H + c, accW[r]);
                unsafeAtomicAdd(p.dWl + g * H + c, accL[r]);
            }
        }
    }
}

}  // namespace

#define MIX_DISPATCH(Hv, HDv, ...)                                                     \
    switch ((Hv) * 100 + (HDv)) {                                                      \
        case 232: { constexpr int HH = 2, HD_ = 32; __VA_ARGS__; break; }              \
        case 264: { constexpr int HH = 2, HD_ = 64; __VA_ARGS__; break; }              \
        case 432: { constexpr int HH = 4, HD_ = 32; __VA_ARGS__; break; }              \
        case 464: { constexpr int HH = 4, HD_ = 64; __VA_ARGS__; break; }              \
        case 832: { constexpr int HH = 8, HD_ = 32; __VA_ARGS__; break; }              \
        case 864: { constexpr int HH = 8, HD_ = 64; __VA_ARGS__; break; }              \
        case 1232: { constexpr int HH = 12, HD_ = 32; __VA_ARGS__; break; }            \
        case 1264: { constexpr int HH = 12, HD_ = 64; __VA_ARGS__; break; }            \
        default: dclip_set_error("attn_mix: no instantiation for H=%d hd=%d", (int)(Hv), (int)(HDv)); return DCLIP_EINVAL; \
    }

extern "C" int dclip_attn_mix_supported(int64_t H, int64_t N, int64_t hd) {
    // width % 128: the key tile's chunk permutation needs D / 8 % 16 == 0
    return (H == 2 || H == 4 || H == 8 || H == 12) && (hd == 32 || hd == 64) && (H * hd) % 128 == 0 && N >= 1 && N <= 128;
}

extern "C" int dclip_attn_mix_fwd(const void* qkv, int64_t ld, const float* Wl, const float* Ww, void* R, float* stats, int64_t B,
                                  int64_t H, int64_t N, int64_t Np, int64_t hd, float scale, void* stream) {
    DCLIP_REQUIRE(qkv && Wl && Ww && R && stats && B > 0, "dclip_attn_mix_fwd: null / empty argument");
    DCLIP_REQUIRE(dclip_attn_mix_supported(H, N, hd), "dclip_attn_mix_fwd: unsupported shape H=%ld N=%ld hd=%ld", (long)H, (long)N, (long)hd);
    DCLIP_REQUIRE(Np % 8 == 0 && Np >= N && ld % 8 == 0 && ((uintptr_t)qkv % 16) == 0 && ((uintptr_t)R % 8) == 0, "dclip_attn_mix_fwd: misaligned buffers");
    const int QT = (int)((N + 15) / 16);
    MixFwd p{(const bf16_t*)qkv, ld, Wl, Ww, (bf16_t*)R, stats, (int)B, (int)N, (int)Np, (int)(H * hd), QT, scale};
    DCLIP_REQUIRE((H * hd) % 128 == 0, "dclip_attn_mix_fwd: width %ld must be a multiple of 128", (long)(H * hd));
    const dim3 grid((unsigned)B);                    // one workgroup per sample, one wave per 16-query tile
    const size_t lds = (size_t)2 * 16 * H * hd * 2;  // double-buffered key block
    const double el = (double)B * H * N * Np;
    TraceScope tr(DCLIP_TRACE_ATTN, 4.0 * B * H * N * N * hd + 8.0 * el * H, 2.0 * el + 4.0 * B * N * H * hd, stream, (int)(B * H), (int)N, (int)hd, 7);
    MIX_DISPATCH(H, hd, hipLaunchKernelGGL((attn_mix_fwd_kernel<HH, HD_>), grid, dim3(QT * 64), lds, (hipStream_t)stream, p, Wl, Ww));
    return dclip_check_launch("dclip_attn_mix_fwd");
}

extern "C" int dclip_attn_mix_bwd(const void* qkv, int64_t ld, const void* dO, int64_t ldo, const float* Wl, const float* Ww,
                                  const float* stats, void* dS, float* dWl, float* dWw, int64_t B, int64_t H, int64_t N, int64_t Np,
                                  int64_t hd, float scale, void* stream) {
    DCLIP_REQUIRE(qkv && dO && Wl && Ww && stats && dS && dWl && dWw && B > 0, "dclip_attn_mix_bwd: null / empty argument");
    DCLIP_REQUIRE(dclip_attn_mix_supported(H, N, hd), "dclip_attn_mix_bwd: unsupported shape H=%ld N=%ld hd=%ld", (long)H, (long)N, (long)hd);
    DCLIP_REQUIRE(Np % 8 == 0 && Np >= N && ld % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)qkv % 16) == 0 && ((uintptr_t)dO % 16) == 0 && ((uintptr_t)dS % 8) == 0,
                  "dclip_attn_mix_bwd: misaligned buffers");
    const int QT = (int)((N + 15) / 16);
    MixBwd p{(const bf16_t*)qkv, ld, (const bf16_t*)dO, ldo, Wl, Ww, stats, (bf16_t*)dS, dWl, dWw, (int)B, (int)N, (int)Np, (int)(H * hd), QT, scale};
    int blocks = (int)((B * QT + 3) / 4);
    if (blocks > 512) blocks = 512;              // persistent: the weight-gradient tiles end in 2 x H x H atomics per wave
    const size_t lds = (size_t)4 * 2 * WG_TILE;
    const double el = (double)B * H * N * Np;
    TraceScope tr(DCLIP_TRACE_ATTN, 8.0 * B * H * N * N * hd + 20.0 * el * H, 2.0 * el + 8.0 * B * N * H * hd, stream, (int)(B * H), (int)N, (int)hd, 8);
    MIX_DISPATCH(H, hd, hipLaunchKernelGGL((attn_mix_bwd_kernel<HH, HD_>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, p, Wl, Ww));
    return dclip_check_launch("dclip_attn_mix_bwd");
}
