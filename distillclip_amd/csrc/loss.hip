// Fused distillation loss, forward + backward (gfx950).
//
//   reference: model/_loss.py:118-202 (LossCalculator.cal_tow_tower_loss / cal_one_tower_loss),
//              model/component/clip_model.py:37-44 (L2-normalise, logits = img @ txt.T, no logit scale),
//              model/loss_component/{out_l1,out_cos,out_kl,out_ce,clip_cos_diff,hard_label,soft_label,logits_mse}.py
//
// Inputs are the four [B,E] f32 embeddings (student/teacher x image/text).  Outputs: 16 scalars (total + raw per-term
// values) and d loss / d student embeddings.  The [B,B] student / teacher logits are never written to HBM: they are
// recomputed as 16x16 tiles on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32) inside "stripe" workgroups:
//   pass A  stripe (direction, 16 rows): row sums of exp((x-1)/tau) for S (tau=1 and tau) and T (tau)  -> stats
//           (direction 1 works on the transposed logits, so its row sums are the column sums of S / T)
//   pass B  stripe: recompute tiles, form dL/dS for the stripe in LDS ([16, B] f32; the teacher tile only lives in
//           registers), then dS_stripe @ other-modality embeddings on MFMA -> gradient rows (plain stores: deterministic)
// Cosine logits are bounded by 1, so softmax uses the fixed maximum 1/tau and row/column sums are plain additions.
#include "common.h"
#include <math.h>
#include <stdlib.h>

namespace {

constexpr int SC_IMG = 0;      // +0 l1  +1 cos  +2 kl  +3 ce        (raw sums, normalised in the total kernel)
constexpr int SC_TXT = 4;
constexpr int SC_POS = 8, SC_NEG = 9, SC_MSE = 10, SC_DIAG = 11, SC_KL0 = 12, SC_KL1 = 13, SC_LSE0 = 14, SC_LSE1 = 15;
constexpr int NSC = 16;
// The scalar accumulators are replicated: every wave adds to the copy picked by its block / wave index, loss_write_total sums
// the copies.  All of them in ONE cache line meant ~2 000 (row kernel) / ~4 000 (stripe kernel) serialised same-line atomics:
// 33 and 67 us of kernels whose work is a few microseconds, on the step's critical path between forward and backward.
constexpr int NREP = 64, SCSTRIDE = 32;                     // 64 copies, one 128-byte line each

struct LossCfg {
    float w_l1, w_cos, w_kl, w_ce;            // tower terms: scale * percent
    float w_cd, w_hl, w_sl, w_mse;            // cross-modal terms: scale * percent
    float tau;
    int two_tower;
};

struct LossArgs {
    const float* s[2]; const float* t[2];     // [0] image tower, [1] text tower (single tower: only [0])
    float* ds[2];
    float* nrm[4];                            // normalised s_img, s_txt, t_img, t_txt  [B,E]
    float* inv[2];                            // 1 / |s|
    float* stats;                             // [zs][6][Bl]: r1S rtS rtT c1S ctS ctT of the owned rows, one partial per column slice
    const float* gstats;                      // row-block mode with hard / soft label: [6][B] statistics of EVERY row (all ranks'
                                              // pass-A results, gathered by the caller); null otherwise
    float* stats_out;                         // row-block pass-A-only call: [6][Bl] slice-summed statistics for that gather
    int zs;                                   // column slices of the stripe kernels (blockIdx.z): fills the chip at B = 512
    float* dsh[2];                            // d loss / d normalised student embedding [zs][B,E] (partials per slice)
    float* scal;                              // [NREP][SCSTRIDE] atomically accumulated (replicated, see NREP)
    float* out;                               // [16] user-visible scalars
    int B, E;                                 // B: number of columns = rows of the (gathered) inputs
    int r0, Bl;                               // the rows this call owns: [r0, r0 + Bl) (whole matrix: 0, B).  Row-block mode:
                                              // data-parallel ranks evaluate their own row block of the global-negative loss
    LossCfg c;
};

__device__ __forceinline__ float* scal_slot(const LossArgs& a) {
    const unsigned r = (blockIdx.x + 3u * blockIdx.y + 5u * blockIdx.z + 7u * (threadIdx.x >> 6)) % NREP;
    return a.scal + r * SCSTRIDE;
}

__device__ void loss_write_total(const LossArgs& a);

// -------------------------------------------------------------------------------------------------------------
// kernel 1: one wave per sample: tower terms (+ their gradients) and the normalised embeddings
// -------------------------------------------------------------------------------------------------------------
template <int NV>   // float4 chunks per lane, E <= 256 * NV
__device__ __forceinline__ void loss_rows_body(const LossArgs& a) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= a.B) return;
    const int E = a.E;
    const int ntow = a.c.two_tower ? 2 : 1;
    const float tw = a.c.two_tower ? 0.5f : 1.f;
    const bool cross = a.c.two_tower && (a.c.w_cd != 0.f || a.c.w_hl != 0.f || a.c.w_sl != 0.f || a.c.w_mse != 0.f);
    const bool own = b >= a.r0 && b < a.r0 + a.Bl;          // rows outside the block only feed the columns (normalised copies)
    const int bl = b - a.r0;
    for (int tow = 0; tow < ntow; ++tow) {
        const float* sp = a.s[tow] + (int64_t)b * E;
        const float* tp = a.t[tow] + (int64_t)b * E;
        float sv[NV * 4], tv[NV * 4];
        float l1 = 0.f, ss = 0.f, tt = 0.f, st = 0.f, ms = -INFINITY, mt = -INFINITY;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            float4 x = {0.f, 0.f, 0.f, 0.f}, y = {0.f, 0.f, 0.f, 0.f};
            if (c < E) { x = *(const float4*)(sp + c); y = *(const float4*)(tp + c); }
            sv[i * 4 + 0] = x.x; sv[i * 4 + 1] = x.y; sv[i * 4 + 2] = x.z; sv[i * 4 + 3] = x.w;
            tv[i * 4 + 0] = y.x; tv[i * 4 + 1] = y.y; tv[i * 4 + 2] = y.z; tv[i * 4 + 3] = y.w;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float s = sv[i * 4 + k], t = tv[i * 4 + k];
                l1 += fabsf(s - t); ss += s * s; tt += t * t; st += s * t;
                if (c < E) { ms = fmaxf(ms, s); mt = fmaxf(mt, t); }
            }
        }
        l1 = wave_sum(l1); ss = wave_sum(ss); tt = wave_sum(tt); st = wave_sum(st);
        // out_cos.py:10-11 (CosineEmbeddingLoss target=+1): eps 1e-12 added to each squared norm
        const float den = sqrtf((ss + 1e-12f) * (tt + 1e-12f));
        const float cosv = st / den;
        // KL / CE need softmax over the feature axis
        float zs_k = 0.f, zt_k = 0.f, zs_c = 0.f, zt_c = 0.f;
        const bool need_sm = a.c.w_kl != 0.f || a.c.w_ce != 0.f;
        const float itau = a.c.w_kl != 0.f ? 1.f / a.c.tau : 1.f;
        if (need_sm) {
            ms = wave_max(ms); mt = wave_max(mt);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < E) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        zs_k += __expf((sv[i * 4 + k] - ms) * itau); zt_k += __expf((tv[i * 4 + k] - mt) * itau);
                        zs_c += __expf(sv[i * 4 + k] - ms); zt_c += __expf(tv[i * 4 + k] - mt);
                    }
                }
            }
            zs_k = wave_sum(zs_k); zt_k = wave_sum(zt_k); zs_c = wave_sum(zs_c); zt_c = wave_sum(zt_c);
        }
        const float lzs_k = logf(zs_k), lzt_k = logf(zt_k), lzs_c = logf(zs_c), lzt_c = logf(zt_c);
        float kl = 0.f, ce = 0.f;
        const float g_l1 = a.c.w_l1 * tw / ((float)a.B * E);
        const float g_cos = a.c.w_cos * tw / a.B;
        const float inv_s = rsqrtf(ss), inv_t = rsqrtf(tt);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < E) {
                float g[4], sn[4], tn[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float s = sv[i * 4 + k], t = tv[i * 4 + k];
                    const float d = s - t;
                    float gr = g_l1 * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
                    gr -= g_cos * (t / den - cosv * s / (ss + 1e-12f));
                    if (a.c.w_kl != 0.f) {
                        const float lps = (s - ms) * itau - lzs_k, lpt = (t - mt) * itau - lzt_k;
                        const float pt = __expf(lpt);
                        kl += pt * (lpt - lps);
                        gr += a.c.w_kl * tw * a.c.tau * (__expf(lps) - pt);
                    }
                    if (a.c.w_ce != 0.f) {
                        const float lps = (s - ms) - lzs_c, pt = __expf((t - mt) - lzt_c);
                        ce -= pt * lps;
                        gr += a.c.w_ce * tw * (__expf(lps) - pt) / a.B;
                    }
                    g[k] = gr; sn[k] = s * inv_s; tn[k] = t * inv_t;
                }
                if (own) *(float4*)(a.ds[tow] + (int64_t)bl * E + c) = float4{g[0], g[1], g[2], g[3]};
                if (cross) {
                    *(float4*)(a.nrm[tow] + (int64_t)b * E + c) = float4{sn[0], sn[1], sn[2], sn[3]};
                    *(float4*)(a.nrm[2 + tow] + (int64_t)b * E + c) = float4{tn[0], tn[1], tn[2], tn[3]};
                }
            }
        }
        kl = wave_sum(kl); ce = wave_sum(ce);
        if (lane == 0 && own) {
            float* sc = scal_slot(a) + (tow ? SC_TXT : SC_IMG);
            unsafeAtomicAdd(sc + 0, l1);
            unsafeAtomicAdd(sc + 1, 1.f - cosv);
            if (a.c.w_kl != 0.f) unsafeAtomicAdd(sc + 2, kl);
            if (a.c.w_ce != 0.f) unsafeAtomicAdd(sc + 3, ce);
            if (cross) a.inv[tow][bl] = inv_s;
        }
    }
}

template <int NV>
__global__ __launch_bounds__(256) void loss_rows_kernel(LossArgs a) {
    loss_rows_body<NV>(a);
}

// -------------------------------------------------------------------------------------------------------------
// stripe kernels.  Tile of X = Xa[rows] . Xb[cols]^T on v_mfma_f32_16x16x4_f32.  The contraction index is permuted
// so that a lane reads float4s: lane (r = l&15, g = l>>4) covers k = 16*o + 4*g + t at MFMA step t.
// C layout: acc[q] = X[row 4*g + q][col l & 15].
// -------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 logits_tile(const float* __restrict__ xa, const float* __restrict__ xb, int E, int lane) {
    const float* pa = xa + (int64_t)(lane & 15) * E + (lane >> 4) * 4;
    const float* pb = xb + (int64_t)(lane & 15) * E + (lane >> 4) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int o = 0; o < E; o += 16) {
        const float4 av = *(const float4*)(pa + o), bv = *(const float4*)(pb + o);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);
    }
    return acc;
}

__device__ __forceinline__ float rowgroup_sum(float v) {   // sum over the 16 lanes that share l >> 4
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    return v;
}

// grid = (ceil(B/16), 2 directions)
__global__ __launch_bounds__(256) void loss_stripe_a_kernel(LossArgs a) {
    __shared__ float red[4][16][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dir = blockIdx.y;
    const int B = a.B, E = a.E;
    const int i0 = a.r0 + blockIdx.x * 16, iend = a.r0 + a.Bl;   // owned rows (global indices)
    const float* sa = a.nrm[dir], *sb = a.nrm[1 - dir];           // student rows / cols
    const float* ta = a.nrm[2 + dir], *tb = a.nrm[3 - dir];       // teacher rows / cols
    const float itau = a.c.w_sl != 0.f ? 1.f / a.c.tau : 1.f;
    const int ia = min(i0 + (lane & 15), B - 1) - (lane & 15);    // clamp the row panel inside the matrix
    float r1[4] = {0, 0, 0, 0}, rs[4] = {0, 0, 0, 0}, rt[4] = {0, 0, 0, 0};
    float pos = 0.f, neg = 0.f, mse = 0.f, diag = 0.f;
    const int ntile = (B + 15) / 16;
    const int z = blockIdx.z, t0 = (int)((int64_t)ntile * z / a.zs), t1 = (int)((int64_t)ntile * (z + 1) / a.zs);   // this slice's column tiles
    for (int jt = t0 + wave; jt < t1; jt += 4) {
        const int j0 = jt * 16;
        const int jb = min(j0 + (lane & 15), B - 1) - (lane & 15);
        const f32x4 S = logits_tile(sa + (int64_t)ia * E, sb + (int64_t)jb * E, E, lane);
        const f32x4 T = logits_tile(ta + (int64_t)ia * E, tb + (int64_t)jb * E, E, lane);
        const int col = j0 + (lane & 15);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = i0 + (lane >> 4) * 4 + q;
            if (row < iend && col < B) {
                const float s = S[q], t = T[q];
                if (a.c.w_hl != 0.f) r1[q] += __expf(s - 1.f);
                if (a.c.w_sl != 0.f) { rs[q] += __expf((s - 1.f) * itau); rt[q] += __expf((t - 1.f) * itau); }
                if (dir == 0) {
                    if (row == col) { pos += fmaxf(t - s, 0.f); diag += s; }
                    else neg += fmaxf(s - t, 0.f);
                    mse += (s - t) * (s - t);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x = rowgroup_sum(r1[q]), y = rowgroup_sum(rs[q]), z = rowgroup_sum(rt[q]);
        if ((lane & 15) == 0) {
            red[wave][(lane >> 4) * 4 + q][0] = x; red[wave][(lane >> 4) * 4 + q][1] = y; red[wave][(lane >> 4) * 4 + q][2] = z;
        }
    }
    if (dir == 0) {
        pos = wave_sum(pos); neg = wave_sum(neg); mse = wave_sum(mse); diag = wave_sum(diag);
        if (lane == 0) {
            float* sc = scal_slot(a);
            unsafeAtomicAdd(sc + SC_POS, pos); unsafeAtomicAdd(sc + SC_NEG, neg);
            unsafeAtomicAdd(sc + SC_MSE, mse); unsafeAtomicAdd(sc + SC_DIAG, diag);
        }
    }
    __syncthreads();
    if (threadIdx.x < 48) {
        const int r = threadIdx.x % 16, k = threadIdx.x / 16;
        if (i0 + r < iend) {
            const float v = (red[0][r][k] + red[1][r][k]) + (red[2][r][k] + red[3][r][k]);
            a.stats[((int64_t)z * 6 + dir * 3 + k) * a.Bl + i0 - a.r0 + r] = v;
        }
    }
}

// normalisation backward of one owned row (local index b) of tower `tow`, by one wave:  x_hat = x / |x|  =>
// dx = (g - x_hat (x_hat . g)) / |x|, added to the tower-term gradient.  g = the column slices' partial rows, added in slice order.
__device__ __forceinline__ void loss_finalize_row(const LossArgs& a, int tow, int b, int lane) {
    const float* xh = a.nrm[tow] + (int64_t)(a.r0 + b) * a.E;
    float* g = a.dsh[tow] + (int64_t)b * a.E;
    const int64_t zstride = (int64_t)a.Bl * a.E;
    float dot = 0.f;
    for (int c = lane * 4; c < a.E; c += 256) {
        float4 y = *(const float4*)(g + c);
        for (int zz = 1; zz < a.zs; ++zz) {                     // slice order: deterministic
            const float4 p = *(const float4*)(g + zz * zstride + c);
            y.x += p.x; y.y += p.y; y.z += p.z; y.w += p.w;
        }
        *(float4*)(g + c) = y;                                  // slice 0 now holds the complete row
        const float4 x = *(const float4*)(xh + c);
        dot += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
    }
    dot = wave_sum(dot);
    const float inv = a.inv[tow][b];
    float* d = a.ds[tow] + (int64_t)b * a.E;
    for (int c = lane * 4; c < a.E; c += 256) {
        const float4 x = *(const float4*)(xh + c), y = *(const float4*)(g + c);
        float4 o = *(const float4*)(d + c);
        o.x += (y.x - x.x * dot) * inv; o.y += (y.y - x.y * dot) * inv;
        o.z += (y.z - x.z * dot) * inv; o.w += (y.w - x.w * dot) * inv;
        *(float4*)(d + c) = o;
    }
}

__global__ __launch_bounds__(256) void loss_stripe_b_kernel(LossArgs a) {
    extern __shared__ __attribute__((aligned(16))) float dsl[];   // [16][B16 + 4]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dir = blockIdx.y;
    const int B = a.B, E = a.E;
    const int ntile = (B + 15) / 16;
    const int ldl = ((ntile + a.zs - 1) / a.zs) * 16 + 4;       // the LDS stripe holds this slice's columns only
    const int i0 = a.r0 + blockIdx.x * 16, iend = a.r0 + a.Bl;   // owned rows (global indices)
    const int Bl = a.Bl;
    const float* sa = a.nrm[dir], *sb = a.nrm[1 - dir];
    const float* ta = a.nrm[2 + dir], *tb = a.nrm[3 - dir];
    const float tau = a.c.tau, itau = a.c.w_sl != 0.f ? 1.f / tau : 1.f;
    // row statistics of this direction, column statistics = row statistics of the other direction
    // (the column statistics exist only when the call owns every row: hard_label / soft_label are not offered in row-block mode)
    const float* rst = a.stats + (int64_t)dir * 3 * Bl;
    const float* cst = a.stats + (int64_t)(1 - dir) * 3 * Bl;
    // the row / column sums arrive as one partial per column slice of pass A: added in slice order (deterministic)
    auto stat = [&](const float* base, int idx) {
        float v = 0.f;
        for (int zz = 0; zz < a.zs; ++zz) v += base[(int64_t)zz * 6 * Bl + idx];
        return v;
    };
    // with gathered statistics: row k of direction d is a.gstats[(d * 3 + k) * B + global index]
    const float* grow = a.gstats ? a.gstats + (int64_t)dir * 3 * B : nullptr;
    const float* gcol = a.gstats ? a.gstats + (int64_t)(1 - dir) * 3 * B : nullptr;
    const int ia = min(i0 + (lane & 15), B - 1) - (lane & 15);
    float rr1[4], rrs[4], rrt[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = min(i0 + (lane >> 4) * 4 + q, iend - 1) - a.r0;
        if (grow) { rr1[q] = grow[a.r0 + row]; rrs[q] = grow[B + a.r0 + row]; rrt[q] = grow[2 * B + a.r0 + row]; }
        else { rr1[q] = stat(rst, row); rrs[q] = stat(rst, Bl + row); rrt[q] = stat(rst, 2 * Bl + row); }
    }
    const int z = blockIdx.z, t0 = (int)((int64_t)ntile * z / a.zs), t1 = (int)((int64_t)ntile * (z + 1) / a.zs);
    const float k_cd_pos = a.c.w_cd / B, k_cd_neg = B > 1 ? a.c.w_cd / ((float)B * (B - 1)) : 0.f;
    const float k_mse = a.c.w_mse * 2.f / ((float)B * B);
    const float k_hl = a.c.w_hl * 0.5f / B;
    const float k_sl = a.c.w_sl * 0.5f * tau;
    float klacc = 0.f;
    for (int jt = t0 + wave; jt < t1; jt += 4) {
        const int j0 = jt * 16;
        const int jb = min(j0 + (lane & 15), B - 1) - (lane & 15);
        const f32x4 S = logits_tile(sa + (int64_t)ia * E, sb + (int64_t)jb * E, E, lane);
        const f32x4 T = logits_tile(ta + (int64_t)ia * E, tb + (int64_t)jb * E, E, lane);
        const int col = j0 + (lane & 15);
        const int cc = min(col, B - 1);
        const bool cstat = a.c.w_hl != 0.f || a.c.w_sl != 0.f;      // only then (and only with every row owned) are they read
        float c1 = 1.f, cs = 1.f, ct = 1.f;
        if (cstat && gcol) { c1 = gcol[cc]; cs = gcol[B + cc]; ct = gcol[2 * B + cc]; }
        else if (cstat) { c1 = stat(cst, cc); cs = stat(cst, Bl + cc); ct = stat(cst, 2 * Bl + cc); }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rl = (lane >> 4) * 4 + q, row = i0 + rl;
            float d = 0.f;
            if (row < iend && col < B) {
                const float s = S[q], t = T[q];
                // cos_diff (clip_cos_diff.py:16-23) and logits_mse: identical in both directions -> one full share
                if (row == col) d -= (t > s) ? k_cd_pos : 0.f;
                else d += (s > t) ? k_cd_neg : 0.f;
                d += k_mse * (s - t);
                if (a.c.w_hl != 0.f) {
                    const float e = __expf(s - 1.f);
                    d += k_hl * (e / rr1[q] + e / c1 - (row == col ? 2.f : 0.f));
                }
                if (a.c.w_sl != 0.f) {
                    const float es = __expf((s - 1.f) * itau), et = __expf((t - 1.f) * itau);
                    d += k_sl * ((es / rrs[q] - et / rrt[q]) + (es / cs - et / ct));
                    // KLDiv(sum) of this direction's rows: p_t (log p_t - log p_s)
                    const float lpt = (t - 1.f) * itau - __logf(rrt[q]), lps = (s - 1.f) * itau - __logf(rrs[q]);
                    klacc += (et / rrt[q]) * (lpt - lps);
                }
            }
            dsl[rl * ldl + col - t0 * 16] = d;
        }
    }
    if (a.c.w_sl != 0.f) {
        klacc = wave_sum(klacc);
        if (lane == 0) unsafeAtomicAdd(scal_slot(a) + (dir ? SC_KL1 : SC_KL0), klacc);
    }
    if (a.c.w_hl != 0.f && z == 0 && wave == 0 && lane < 16 && i0 + lane < iend)
        unsafeAtomicAdd(scal_slot(a) + (dir ? SC_LSE1 : SC_LSE0), __logf(grow ? grow[i0 + lane] : stat(rst, i0 - a.r0 + lane)) + 1.f);
    __syncthreads();
    // gradient rows: G[16, E] = dS_stripe[16, B] @ Y[B, E],  Y = normalised student embedding of the other modality
    // (this slice's columns only: the slices' partial rows are added by the stripe's last-arriving slice, loss_finalize_row)
    const float* Y = sb;
    float* G = a.dsh[dir] + (int64_t)z * Bl * E;
    const float* arow = dsl + (lane & 15) * ldl + (lane >> 4) * 4;
    for (int e0 = wave * 16; e0 < E; e0 += 64) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* yb = Y + e0 + (lane & 15);
        for (int o = 0; o < (t1 - t0) * 16; o += 16) {
            const float4 av = *(const float4*)(arow + o);
            const int k0 = t0 * 16 + o + (lane >> 4) * 4;
            // rows beyond B carry zero weights in dsl; clamp the address only
            const float b0 = yb[(int64_t)min(k0 + 0, B - 1) * E], b1 = yb[(int64_t)min(k0 + 1, B - 1) * E];
            const float b2 = yb[(int64_t)min(k0 + 2, B - 1) * E], b3 = yb[(int64_t)min(k0 + 3, B - 1) * E];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b3, acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = i0 + (lane >> 4) * 4 + q;
            if (row < iend) G[(int64_t)(row - a.r0) * E + e0 + (lane & 15)] = acc[q];
        }
    }
}

// pass-A-only call of the row-block mode: the owned rows' statistics, slices added in order, for the caller's all-gather
__global__ void loss_stats_out_kernel(LossArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 6 * a.Bl) return;
    float v = 0.f;
    for (int zz = 0; zz < a.zs; ++zz) v += a.stats[(int64_t)zz * 6 * a.Bl + i];
    a.stats_out[i] = v;
}

// out[0] total ; [1..4] image l1,cos,kl,ce ; [5..8] text ; [9..12] cos_diff, hard_label, soft_label, logits_mse (raw)
// (run by every thread of one workgroup: the extra workgroup of loss_finalize_kernel, or loss_total_kernel)
__device__ void loss_write_total(const LossArgs& a) {
    __shared__ float tot[NSC];
    if (threadIdx.x < NSC) {
        float x = 0.f;
        for (int r = 0; r < NREP; ++r) x += a.scal[r * SCSTRIDE + threadIdx.x];
        tot[threadIdx.x] = x;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const float B = (float)a.B, E = (float)a.E;
    const LossCfg& c = a.c;
    float o[16];
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
    const int ntow = c.two_tower ? 2 : 1;
    float total = 0.f;
    for (int t = 0; t < ntow; ++t) {
        const float* sc = tot + (t ? SC_TXT : SC_IMG);
        float* ot = o + 1 + 4 * t;
        ot[0] = sc[0] / (B * E); ot[1] = sc[1] / B; ot[2] = sc[2] * c.tau * c.tau; ot[3] = sc[3] / B;
        const float tl = c.w_l1 * ot[0] + c.w_cos * ot[1] + c.w_kl * ot[2] + c.w_ce * ot[3];
        total += (c.two_tower ? 0.5f : 1.f) * tl;
    }
    if (c.two_tower) {
        const float* s = tot;
        // B = 1 with cos_diff switched on: mean over no negatives = 0 / 0 = NaN, as the reference (clip_cos_diff.py:16-23);
        // with the term off the raw value must stay finite (0 * NaN would poison the total)
        o[9] = s[SC_POS] / B + ((a.B > 1 || a.c.w_cd != 0.f) ? s[SC_NEG] / (B * (B - 1.f)) : 0.f);
        o[10] = c.w_hl != 0.f ? 0.5f * ((s[SC_LSE0] - s[SC_DIAG]) / B + (s[SC_LSE1] - s[SC_DIAG]) / B) : 0.f;
        o[11] = 0.5f * (s[SC_KL0] + s[SC_KL1]) * c.tau * c.tau;
        o[12] = s[SC_MSE] / (B * B);
        total += c.w_cd * o[9] + c.w_hl * o[10] + c.w_sl * o[11] + c.w_mse * o[12];
    }
    o[0] = total;
    for (int i = 0; i < 16; ++i) a.out[i] = o[i];
}

// default path: the normalisation backward of every owned row (one wave each) as a launch of its own; the extra last workgroup writes
// the 16 scalars (they depend on the earlier launches only, not on this one)
__global__ __launch_bounds__(256) void loss_finalize_kernel(LossArgs a) {
    if (blockIdx.x == gridDim.x - 1) { loss_write_total(a); return; }
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);          // owned row (local index)
    if (b >= a.Bl) return;
    loss_finalize_row(a, 0, b, lane);
    loss_finalize_row(a, 1, b, lane);
}
__global__ __launch_bounds__(64) void loss_total_kernel(LossArgs a) { loss_write_total(a); }

inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }
// arrival counters behind the scalar accumulators: one for the call, one per (direction, 16-row stripe)

}  // namespace

extern "C" size_t dclip_distill_loss_workspace(int64_t B, int64_t E) {
    const size_t be = align_up((size_t)B * E * sizeof(float));
    const size_t zs = 8;                                     // upper bound of loss_slices() for any row block
    return (4 + 2 * zs) * be + align_up((size_t)2 * B * 4) + align_up(zs * 6 * B * 4) + align_up((size_t)NREP * SCSTRIDE * 4);
}

namespace {

int run_distill_loss(const float* s_img, const float* t_img, const float* s_txt, const float* t_txt, int64_t B, int64_t E,
                     int64_t row0, int64_t rows, const float* cfg, float* out_scalars, float* d_s_img, float* d_s_txt,
                     void* workspace, size_t ws_bytes, void* stream, const char* who, const float* gathered_stats = nullptr,
                     float* stats_out = nullptr) {
    DCLIP_REQUIRE(cfg && out_scalars && workspace, "%s: null argument", who);
    LossArgs a;
    a.c.w_l1 = cfg[0]; a.c.w_cos = cfg[1]; a.c.w_kl = cfg[2]; a.c.w_ce = cfg[3];
    a.c.w_cd = cfg[4]; a.c.w_hl = cfg[5]; a.c.w_sl = cfg[6]; a.c.w_mse = cfg[7];
    a.c.tau = cfg[8]; a.c.two_tower = cfg[9] != 0.f;
    DCLIP_REQUIRE(s_img && t_img && d_s_img, "%s: tower 0 pointers are required", who);
    DCLIP_REQUIRE(!a.c.two_tower || (s_txt && t_txt && d_s_txt), "%s: two-tower mode needs both towers", who);
    DCLIP_REQUIRE(B > 0 && E > 0 && E % 16 == 0 && E <= 1024, "%s: need E %% 16 == 0, E <= 1024 (E=%ld)", who, (long)E);
    DCLIP_REQUIRE(B <= 4096, "%s: B <= 4096 (stripe LDS budget), got %ld", who, (long)B);
    DCLIP_REQUIRE(row0 >= 0 && rows > 0 && row0 + rows <= B, "%s: row block [%ld, %ld) outside [0, %ld)", who, (long)row0, (long)(row0 + rows), (long)B);
    DCLIP_REQUIRE(rows == B || (a.c.w_hl == 0.f && a.c.w_sl == 0.f) || gathered_stats || stats_out,
                  "%s: hard_label / soft_label need the statistics of every row: call once with stats_out, gather, call again with them", who);
    a.gstats = gathered_stats; a.stats_out = stats_out;
    DCLIP_REQUIRE((a.c.w_kl == 0.f && a.c.w_sl == 0.f) || a.c.tau > 0.f, "%s: KL terms need temperature > 0", who);
    DCLIP_REQUIRE(ws_bytes >= dclip_distill_loss_workspace(B, E), "%s: workspace too small", who);
    DCLIP_REQUIRE(((uintptr_t)workspace % 256) == 0, "%s: workspace must be 256-byte aligned", who);
    a.s[0] = s_img; a.s[1] = s_txt; a.t[0] = t_img; a.t[1] = t_txt; a.ds[0] = d_s_img; a.ds[1] = d_s_txt;
    a.B = (int)B; a.E = (int)E; a.r0 = (int)row0; a.Bl = (int)rows; a.out = out_scalars;
    char* w = (char*)workspace;
    const size_t be = align_up((size_t)B * E * sizeof(float));
    for (int i = 0; i < 4; ++i) { a.nrm[i] = (float*)w; w += be; }
    // column slices: enough workgroups for the chip from the owned row stripes
    {
        const int rtile = (int)((rows + 15) / 16), ctile = (int)((B + 15) / 16);
        int zs = 256 / (2 * rtile);
        zs = zs < 1 ? 1 : (zs > 8 ? 8 : zs);
        zs = zs > ctile ? ctile : zs;
        while (zs < 8 && (size_t)16 * (((ctile + zs - 1) / zs) * 16 + 4) * sizeof(float) > 144 * 1024) ++zs;   // LDS stripe budget
        a.zs = zs;
    }
    for (int i = 0; i < 2; ++i) { a.dsh[i] = (float*)w; w += (size_t)a.zs * be; }
    a.inv[0] = (float*)w; a.inv[1] = a.inv[0] + rows; w += align_up((size_t)2 * B * 4);
    a.stats = (float*)w; w += align_up((size_t)a.zs * 6 * B * 4);
    a.scal = (float*)w;
    const bool cross = a.c.two_tower && (a.c.w_cd != 0.f || a.c.w_hl != 0.f || a.c.w_sl != 0.f || a.c.w_mse != 0.f);
    hipStream_t st = (hipStream_t)stream;
    // algorithmic HBM bytes: read 4*B*E*4 + write 2*rows*E*4 ; the logits contribute none
    TraceScope tr(DCLIP_TRACE_LOSS, 0.0, (a.c.two_tower ? 2.0 : 1.0) * (2.0 * (double)B + (double)rows) * E * 4.0, stream);
    // launches of a call: this fill (the replicated scalar accumulators), rows, and with cross-modal terms stripe A, stripe B
    // and the normalisation backward, whose extra workgroup writes the 16 scalars (round 3: a fifth launch)
    if (hipMemsetAsync(a.scal, 0, align_up((size_t)NREP * SCSTRIDE * sizeof(float)), st) != hipSuccess) {
        dclip_set_error("%s: memset failed", who);
        return DCLIP_ELAUNCH;
    }
    const dim3 allrows((unsigned)((B + 3) / 4));
    const int nv = (int)((E + 255) / 256);
    switch (nv) {
        case 1: hipLaunchKernelGGL((loss_rows_kernel<1>), allrows, dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL((loss_rows_kernel<2>), allrows, dim3(256), 0, st, a); break;
        case 3: hipLaunchKernelGGL((loss_rows_kernel<3>), allrows, dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL((loss_rows_kernel<4>), allrows, dim3(256), 0, st, a); break;
    }
    if (cross) {
        const dim3 grid((unsigned)((rows + 15) / 16), 2, (unsigned)a.zs);
        hipLaunchKernelGGL(loss_stripe_a_kernel, grid, dim3(256), 0, st, a);
        if (stats_out && !gathered_stats) {      // pass A only: the caller gathers the statistics of all row blocks first
            hipLaunchKernelGGL(loss_stats_out_kernel, dim3((unsigned)((6 * rows + 255) / 256)), dim3(256), 0, st, a);
            return dclip_check_launch(who);
        }
        const size_t lds = (size_t)16 * (((((B + 15) / 16) + a.zs - 1) / a.zs) * 16 + 4) * sizeof(float);
        DCLIP_REQUIRE(lds <= 160 * 1024, "%s: stripe does not fit LDS", who);
        hipLaunchKernelGGL(loss_stripe_b_kernel, grid, dim3(256), lds, st, a);
        hipLaunchKernelGGL(loss_finalize_kernel, dim3((unsigned)((rows + 3) / 4) + 1), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL(loss_total_kernel, dim3(1), dim3(64), 0, st, a);
    }
    return dclip_check_launch(who);
}

}  // namespace

extern "C" int dclip_distill_loss(const float* s_img, const float* t_img, const float* s_txt, const float* t_txt, int64_t B,
                                  int64_t E, const float* cfg, float* out_scalars, float* d_s_img, float* d_s_txt,
                                  void* workspace, size_t ws_bytes, void* stream) {
    return run_distill_loss(s_img, t_img, s_txt, t_txt, B, E, 0, B, cfg, out_scalars, d_s_img, d_s_txt, workspace, ws_bytes, stream,
                            "dclip_distill_loss");
}

extern "C" int dclip_distill_loss_rows(const float* s_img, const float* t_img, const float* s_txt, const float* t_txt, int64_t B,
                                       int64_t E, int64_t row0, int64_t rows, const float* cfg, float* out_scalars,
                                       float* d_s_img, float* d_s_txt, const float* gathered_stats, float* stats_out,
                                       void* workspace, size_t ws_bytes, void* stream) {
    return run_distill_loss(s_img, t_img, s_txt, t_txt, B, E, row0, rows, cfg, out_scalars, d_s_img, d_s_txt, workspace, ws_bytes,
                            stream, "dclip_distill_loss_rows", gathered_stats, stats_out);
}

// feature MSE (hidden_rep_mse / embedding_mse terms: hidden_mse.py:9-17, embed_mse.py:9-10):
//   acc[0] += coef * mean((s - t)^2) ; ds_acc += coef * 2 (s - t) / n      (ds_acc may be null)
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ s, const float* __restrict__ t, int64_t n, float coef,
                                                  float* __restrict__ acc, float* __restrict__ ds_acc) {
    __shared__ float part[4];
    float sum = 0.f;
    const float k = coef * 2.f / (float)n;
    // four independent 16-byte pieces per thread and trip (their loads are all in flight before the first is consumed), one atomic per BLOCK
    const int64_t stride = (int64_t)gridDim.x * 1024;
    for (int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i0 < n; i0 += 4 * stride) {
        float4 x[4], y[4], g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < n) {
                x[u] = *(const float4*)(s + i); y[u] = *(const float4*)(t + i);
                if (ds_acc) g[u] = *(const float4*)(ds_acc + i);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < n) {
                const float a0 = x[u].x - y[u].x, a1 = x[u].y - y[u].y, a2 = x[u].z - y[u].z, a3 = x[u].w - y[u].w;
                sum += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                if (ds_acc) {
                    g[u].x += k * a0; g[u].y += k * a1; g[u].z += k * a2; g[u].w += k * a3;
                    *(float4*)(ds_acc + i) = g[u];
                }
            }
        }
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(acc, ((part[0] + part[1]) + (part[2] + part[3])) * coef / (float)n);
}

extern "C" int dclip_feature_mse(const float* s, const float* t, int64_t n, float coef, float* loss_acc, float* ds_acc,
                                 void* stream) {
    DCLIP_REQUIRE(s && t && loss_acc && n > 0 && n % 4 == 0, "dclip_feature_mse: bad argument (n %% 4 == 0 required)");
    int64_t blocks = (n / 16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(mse_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, s, t, n, coef, loss_acc, ds_acc);
    return dclip_check_launch("dclip_feature_mse");
}
