// Head-mixed student attention, score stage, with BOTH head mixes on the matrix pipe and no lane movement.
//
//   reference: model/component/weight_share_model.py:101-125 (MiniAttention.forward)
//       S_h = scale * Q_h K_h^T ; A = conv_l(S) (1x1 conv over the head channel) ; P = softmax_j(A) ; R = conv_w(P)
//
// This header is the per-WAVE algorithm.  It is compiled twice: by hipcc for gfx950 (attention_mix.hip) and by the host compiler
// against a 64-lane emulation of the wave collectives (tools/emu/) that checks the lane / register index maps on the CPU.
//
// One wave owns (sample b, 16 query rows); lane = (c, g4) = (lane & 15, lane >> 4), c <-> query i0 + c.  Keys are walked four
// at a time ("quad" j0 .. j0 + 3).  v_mfma_f32_16x16x32 computes D[row][col] += sum_k A[row][k] B[k][col] with
//   A fragment of lane l = A[row = l & 15][k = 8 (l >> 4) + 0..7],  B fragment = B[k = 8 (l >> 4) + 0..7][col = l & 15],
//   accumulator register r of lane l = D[row = 4 (l >> 4) + r][col = l & 15].
//
// (1) BLOCK-DIAGONAL scores.  For a set s of four heads, rows of the MFMA are (head-in-set hh, key-in-quad jk) = 4 hh + jk and the
//     k slots are (lane group grp, 8 channels): A[(hh, jk)][(grp, d8)] = [grp == hh] K_{4s+hh}[j0 + jk][8 ci + d8],
//     B[(grp, d8)][c] = Q_{4s+grp}[i0 + c][8 ci + d8].  Summed over the HD / 8 channel chunks ci, accumulator register r of lane
//     (c, g4) is S_{4s+g4}[i0 + c][j0 + r]: lane group g4 ends up with head 4s + g4.  Three quarters of the A operand are zeros
//     (the 16 lanes with (c >> 2) == g4 carry a key row each) -- four times the MFMA work of a dense tile, on a pipe that is idle.
// (2) MIX 1 from that layout.  For key j0 + r, the packed values {S-acc[s][r] : s} of lane (c, g4) ARE a B fragment whose k slot
//     (g4, jj) means head 4 jj + g4; with the mix matrix as the A operand (columns permuted the same way) one MFMA per 16 output
//     heads gives A[g][c] for that key: accumulator register r' of lane (c, g4) <-> output head 16 t + 4 g4 + r'.
// (3) Softmax statistics are per (output head, query) = per accumulator register of a lane: running maximum / sum over the keys are
//     plain register updates, no cross-lane step at all.  -lse enters as the initial accumulator, so P = exp2(mfma result).
// (4) MIX 2 (and the adjoint mixes of the backward) contract over the ROW index of an accumulator tile, so the packed accumulator
//     registers are the next B fragment as they stand (k slot (g4, jj) <-> head 16 (jj >> 2) + 4 g4 + (jj & 3)).
//
// Precision: q, k, v, dO are bf16 (exact products, f32 accumulation).  The forward mixes run on f16 operands (11-bit mantissa:
// the precision the reference's fp16 autocast gives conv_l / conv_w); the backward's gradient-side operands are bf16 (range).
#pragma once
#include <type_traits>

namespace amix {

template <int H_, int HD_>
struct Cfg {
    static constexpr int H = H_, HD = HD_;
    static constexpr int D = H * HD;
    static constexpr int NS = (H + 3) / 4;       // sets of four heads
    static constexpr int RT = (H + 15) / 16;     // 16-row tiles of a mixed tensor
    static constexpr int NC = HD / 8;            // 8-channel chunks of a head row
    static constexpr int HP = 16 * RT;           // padded head count of the weight-gradient tiles
};

template <class T8> struct elem_of;
template <> struct elem_of<bf16x8> { typedef bf16_t type; };
template <> struct elem_of<f16x8> { typedef _Float16 type; };

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float RESCALE_THR = 24.f;              // lazy rescale of the running softmax reference (log2 units)
constexpr float P_OFF = 1e30f;                   // "log-sum-exp" of rows that must come out as P = 0

struct FwdArgs {
    const bf16_t* qkv; long ld;                  // [B*N, 3D]: q | k | v, head h at column h * HD
    const float* Wl; const float* Ww;            // [H, H] conv_l / conv_w weights (f32 masters)
    bf16_t* R;                                   // mixed probabilities, quad-blocked [B, H, Np / 4, N, 4], pad columns zero
    float* stats;                                // [B, H, N] log-sum-exp (natural log) of row (b, h, i) of A
    int B, N, Np, QT;
    float scale;
    unsigned long long* stamps;                  // diagnostics (nullable): 12 cycle counts per (sample, tile), see fwd_item
};

struct BwdArgs {
    const bf16_t* qkv; long ld;
    const bf16_t* dO; long ldo;                  // [B*N, D] gradient of ctx
    const float* Wl; const float* Ww;
    const float* stats;
    bf16_t* dS;                                  // gradient of the scaled pre-mix scores, quad-blocked like R, pad columns zero
    float* partial;                              // [workgroups][2][HP][HP] weight-gradient partial sums (dW_l, dW_w)
    float* delta;                                // [B, H, N] sum_j P dP of every (head, query): pass A writes, pass B reads
    int B, N, Np, QT;
    float scale;
    unsigned long long* stamps;                  // diagnostics (nullable): 8 cycle counts per (sample, tile), see bwd_item
};

// ---- weight operands -------------------------------------------------------------------------------------------------------
// A fragment of an [H, H] mix matrix for the products above.  elem(out, in) returns M[out][in].
//   SLAYOUT: k slot (grp, jj) <-> input head 4 jj + grp            (B operand packed from block-diagonal score accumulators)
//   else   : k slot (grp, jj) <-> input head 16 (jj >> 2) + 4 grp + (jj & 3)   (B operand packed from a mixed accumulator tile)
template <class C, bool SLAYOUT, class T8, class F>
DEVFN T8 weight_frag(int lane, int t, F elem) {
    const int row = 16 * t + (lane & 15), grp = lane >> 4;
    T8 f;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int in = SLAYOUT ? 4 * jj + grp : 16 * (jj >> 2) + 4 * grp + (jj & 3);
        const bool ok = row < C::H && in < C::H && (!SLAYOUT || jj < C::NS);
        const float v = elem(row < C::H ? row : C::H - 1, in < C::H ? in : C::H - 1);      // unconditional: the loads overlap
        f[jj] = (typename elem_of<T8>::type)(ok ? v : 0.f);
    }
    return f;
}

// ---- block-diagonal scores ---------------------------------------------------------------------------------------------------
// Column-entity fragments (queries here): lane (c, g4) holds rows of head 4s + g4 of token `tok`, chunk ci.
// PIN: keep the fragments in the accumulator half of the register file (they are MFMA-only operands, which may be AGPRs): the
// one-wave-per-SIMD backward kernels then keep their 256 VGPRs for what the VALU touches
template <class C, bool PIN>
DEVFN void load_col_frags(const bf16_t* mat, long ld, int tok, bool tok_ok, int lane, bf16x8 (&f)[C::NS][C::NC]) {
    const int g4 = lane >> 4;
    // unconditional loads (clamped token / head) so that all of them are in flight together; lanes without a row are zeroed after
    const bf16_t* row = mat + (long)(tok_ok ? tok : 0) * ld;
#pragma unroll
    for (int s = 0; s < C::NS; ++s) {
        const bool head_ok = (C::H % 4 == 0) || 4 * s + g4 < C::H;
        const bf16_t* src = row + (head_ok ? 4 * s + g4 : 0) * C::HD;
#pragma unroll
        for (int ci = 0; ci < C::NC; ++ci) f[s][ci] = *(const bf16x8*)(src + 8 * ci);
    }
#pragma unroll
    for (int s = 0; s < C::NS; ++s) {
        const bool ok = tok_ok && ((C::H % 4 == 0) || 4 * s + g4 < C::H);
#pragma unroll
        for (int ci = 0; ci < C::NC; ++ci) {
            const bf16x8 z = {};
            f[s][ci] = ok ? f[s][ci] : z;
            if (PIN) hw::pin_acc(f[s][ci]);
        }
    }
}

// ---- row matrices (keys, values) through a wave-private LDS ring ----------------------------------------------------------------
// A quad of token rows (4 x D bf16) is one stage, filled by LDS-DMA (global_load_lds, 16 B per lane, no VGPR staging) one quad
// ahead of its use.  LDS-DMA writes lane-linear, so the layout is chosen on the SOURCE side: position (row r, chunk cp) of a stage
// holds source chunk cp ^ r of row r0 + r, which puts the 4 rows of one head chunk on 4 different bank groups for the
// ds_read_b128 fragment reads below.
template <class C>
struct Ring {
    static constexpr int ROWB = C::D * 2;            // bytes of one token row (all heads)
    static constexpr int STAGE = 4 * ROWB;           // one quad
    static constexpr int NINST = STAGE / 1024;       // LDS-DMA wave-instructions per stage (D % 128 == 0)
    static constexpr int CPR = C::D / 8;             // 16-byte chunks per row
};

template <class C>
DEVFN void stage_quad(const bf16_t* rows, long ld, int r0, int nrows, char* stage, int lane) {
    constexpr int CPR = Ring<C>::CPR;
    const char* quad = (const char*)(rows + (long)r0 * ld);                      // wave-uniform base, 32-bit lane offsets
    const int last = nrows - 1 - r0;                                              // rows past the end re-read the last row: the keys
#pragma unroll                                                                    // they stand for are masked where they are consumed
    for (int k = 0; k < Ring<C>::NINST; ++k) {
        const int pos = k * 64 + lane;
        const int r = pos / CPR, cp = pos - r * CPR;
        const int rr = r < last ? r : last;
        const unsigned off = (unsigned)rr * (unsigned)(ld * 2) + (unsigned)((cp ^ r) << 4);
        hw::dma16(quad + off, stage + k * 1024);
    }
}

// acc[s][r] (lane (c, g4)) = sum_d Row_{4s+g4}[r0 + r][d] * Col_{4s+g4}[c][d], rows from a staged quad.  The 16 lanes with
// (c >> 2) == g4 carry one row each (head-in-set g4, key-in-quad c & 3); the other 48 read the wave's zero block.
template <class C>
DEVFN void bd_scores(const char* stage, const char* zeros, int lane, const bf16x8 (&colf)[C::NS][C::NC], f32x4 (&acc)[C::NS]) {
    constexpr int NS = C::NS, NC = C::NC;
    const int c = lane & 15, g4 = lane >> 4;
    const int row = c & 3;
    const bool lane_ok = (c >> 2) == g4;
    const char* base = lane_ok ? stage + row * Ring<C>::ROWB + 16 * g4 * NC : zeros;
    const int x = lane_ok ? row : 0;
    // Schedule (one or two waves per SIMD: the wave has to hide its own LDS latency).  The sets are cut into groups of 8 fragments;
    // the reads of group g + 1 are issued before the MFMAs of group g, and inside a group the MFMAs go chunk-major, so that two
    // MFMAs on the same accumulator are GS issues apart.  sched_fence() pins that order for the compiler.
    constexpr int GS = (8 / NC) < NS ? (8 / NC) : NS;          // sets per group
    constexpr int NG = (NS + GS - 1) / GS;
    constexpr int GF = GS * NC;                                // fragments per group
    bf16x8 kf[2][GF];
    // NC lane addresses (chunk ci of the lane's row, swizzled), the set index is an immediate offset of the read.  The fragment
    // reads are hidden from the compiler's LDS-DMA alias tracking (hw::lds_read_frag), which otherwise puts a vmcnt(0) -- the full
    // latency of the request just made for the NEXT quad -- in front of them; hw::lds_wait_frags is the counted lgkmcnt wait that
    // makes a group's registers valid (LDS returns in order: N later reads may stay outstanding)
    const char* addr[NC];
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) addr[ci] = base + 16 * (ci ^ x);
    auto fetch_set = [&](auto sc, int buf, int u) {           // set index as a type: it becomes the immediate offset of the reads
        constexpr int sraw = decltype(sc)::value;
        constexpr int s = sraw < NS ? sraw : 0;
        const bool head_ok = sraw < NS && ((C::H % 4 == 0) || 4 * s + g4 < C::H);
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) {
            if ((C::H % 4 == 0) && sraw < NS) kf[buf][u * NC + ci] = hw::lds_read_frag<64 * s * NC>(addr[ci]);
            else kf[buf][u * NC + ci] = hw::lds_read_frag<0>(head_ok ? addr[ci] + 64 * s * NC : zeros);
        }
    };
    auto fetch = [&](auto gc, int buf) {
        constexpr int g = decltype(gc)::value;
        fetch_set(std::integral_constant<int, g * GS>{}, buf, 0);
        if constexpr (GS > 1) fetch_set(std::integral_constant<int, g * GS + 1>{}, buf, 1);
        static_assert(GS <= 2, "a group is one or two sets");
    };
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto body = [&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g + 1 < NG) { fetch(std::integral_constant<int, g + 1>{}, (g + 1) & 1); hw::lds_wait_frags<GF>(kf[g & 1]); }
        else hw::lds_wait_frags<0>(kf[g & 1]);
#pragma unroll
        for (int ci = 0; ci < NC; ++ci)
#pragma unroll
            for (int u = 0; u < GS; ++u) {
                const int s = g * GS + u;
                if (s < NS) acc[s] = hw::mfma_bf16(kf[g & 1][u * NC + ci], colf[s][ci], acc[s]);
            }
        hw::sched_fence();
    };
    fetch(std::integral_constant<int, 0>{}, 0);
    body(std::integral_constant<int, 0>{});
    if constexpr (NG > 1) body(std::integral_constant<int, 1>{});
    if constexpr (NG > 2) body(std::integral_constant<int, 2>{});
    if constexpr (NG > 3) body(std::integral_constant<int, 3>{});
    static_assert(NG <= 4, "at most 4 fragment groups (NS <= 8)");
}

// NMAT row matrices walked quad by quad through a double-buffered ring: stage (2 m + parity) belongs to matrix m.
template <class C, int NMAT>
struct RowStream {
    const bf16_t* rows[NMAT];
    long ld; int nrows;
    char* ring;
    int lane, par;
    DEVMEM void issue(int q, int pp) const {
#pragma unroll
        for (int m = 0; m < NMAT; ++m) stage_quad<C>(rows[m], ld, 4 * q, nrows, ring + (2 * m + pp) * Ring<C>::STAGE, lane);
    }
    // quad `next` is requested into the other parity, then everything older than that request has landed: the current quad
    DEVMEM void advance(int next) {
        issue(next, par ^ 1);
        hw::dma_wait<NMAT * Ring<C>::NINST>();
    }
    DEVMEM const char* stage(int m) const { return ring + (2 * m + par) * Ring<C>::STAGE; }
    DEVMEM void done() { par ^= 1; }
};

// B fragment of the first-kind mix for instance r of the quad: k slot (g4, jj) <- acc[jj][r]
template <class C, class T8>
DEVFN T8 pack_s(const f32x4 (&acc)[C::NS], int r) {
    T8 f = {};
#pragma unroll
    for (int s = 0; s < C::NS; ++s) f[s] = (typename elem_of<T8>::type)acc[s][r];
    return f;
}
// B fragment of the second-kind mix: k slot (g4, jj) <- x[jj >> 2][jj & 3]
template <class C, class T8>
DEVFN T8 pack_a(const f32x4 (&x)[C::RT]) {
    T8 f = {};
#pragma unroll
    for (int t = 0; t < C::RT; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) f[4 * t + k] = (typename elem_of<T8>::type)x[t][k];
    return f;
}

// ---- forward ---------------------------------------------------------------------------------------------------------------
template <class C>
struct FwdWeights {
    f16x8 wl[C::RT];     // log2(e) * scale * W_l, first-kind layout
    f16x8 ww[C::RT];     // W_w, second-kind layout
};

template <class C>
DEVFN void fwd_load_weights(const FwdArgs& p, int lane, FwdWeights<C>& w) {
    const float* Wl = p.Wl; const float* Ww = p.Ww;
    const float sc = p.scale * LOG2E;
#pragma unroll
    for (int t = 0; t < C::RT; ++t) {
        w.wl[t] = weight_frag<C, true, f16x8>(lane, t, [&](int g, int h) { return Wl[g * C::H + h] * sc; });
        w.ww[t] = weight_frag<C, false, f16x8>(lane, t, [&](int g, int h) { return Ww[g * C::H + h]; });
    }
}

template <class C>
constexpr int fwd_lds_per_wave() { return Ring<C>::ROWB + 2 * Ring<C>::STAGE; }        // zero block + the key ring

// One (sample, 16-query tile): pass 1 = softmax statistics, pass 2 = P, R.  R is written as 16-byte groups of 8 keys.
// lds: this wave's region (fwd_lds_per_wave bytes), its first ROWB bytes already zero.
template <class C>
DEVFN void fwd_item(const FwdArgs& p, int b, int it, int lane, const FwdWeights<C>& w, char* lds) {
    constexpr int NS = C::NS, RT = C::RT, H = C::H, D = C::D;
    const int c = lane & 15, g4 = lane >> 4;
    const int N = p.N;
    const int i = it * 16 + c;
    const bool iok = i < N;
    const bf16_t* base = p.qkv + (long)b * N * p.ld;
    const unsigned long long tq0 = p.stamps ? hw::clock() : 0;
    bf16x8 qf[NS][C::NC];
    load_col_frags<C, false>(base, p.ld, i, iok, lane, qf);
    const char* zeros = lds;
    RowStream<C, 1> ks{{base + D}, p.ld, N, lds + Ring<C>::ROWB, lane, 0};
    const int nq = (N + 3) >> 2;

    // diagnostics: cycles spent waiting for the ring / in the score MFMAs / in the per-key stage, summed over the quads
    const bool st = p.stamps != nullptr;
    unsigned long long t_wait = 0, t_s = 0, t_i = 0, t0 = st ? hw::clock() : 0, tq = 0;
    if (st && lane == 0) p.stamps[12 * ((long)b * p.QT + it) + 9] = t0 - tq0;          // query fragments

    // ---- pass 1: reference value from the first quad, then running sums with a lazy rescale ------------------------------------
    f32x4 m[RT], l[RT];
    ks.issue(0, 0);
    hw::dma_wait<0>();
    {
        f32x4 sa[NS];
        bd_scores<C>(ks.stage(0), zeros, lane, qf, sa);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r < N) {
                const f16x8 pk = pack_s<C, f16x8>(sa, r);
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    const f32x4 a = hw::mfma_f16(w.wl[t], pk, z);
#pragma unroll
                    for (int k = 0; k < 4; ++k) m[t][k] = r == 0 ? a[k] : fmaxf(m[t][k], a[k]);
                }
                hw::mfma_src_guard();            // (once per tile: the operand must outlive its MFMAs, see hw::keep_alive)
            }
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) l[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // TAIL (compile time) = the quad may reach past N: only that variant carries the per-key validity branches
    auto stats_quad = [&](auto tail_c, int q) {
        constexpr bool TAIL = decltype(tail_c)::value;
        if (st) tq = hw::clock();
        ks.advance(q + 1 < nq ? q + 1 : 0);          // the last request is quad 0 of pass 2
        if (st) { const unsigned long long t = hw::clock(); t_wait += t - tq; tq = t; }
        f32x4 sa[NS];
        bd_scores<C>(ks.stage(0), zeros, lane, qf, sa);
        ks.done();
        if (st) { const unsigned long long t = hw::clock(); t_s += t - tq; tq = t; }
        // the quad's scores minus the reference first, the decision to move the reference next, the exponentials last: nothing
        // is ever exponentiated against a reference it exceeds by more than RESCALE_THR
        f32x4 av[4][RT], mx[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) mx[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        f16x8 pk[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pk[r] = pack_s<C, f16x8>(sa, r);
#pragma unroll
            for (int t = 0; t < RT; ++t) av[r][t] = hw::mfma_f16(w.wl[t], pk[r], -m[t]);   // log2-domain score minus the reference
        }
        hw::sched_fence();
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (TAIL && 4 * q + r >= N) av[r][t][k] = -P_OFF;                     // keys beyond N: exp2 -> 0, never the maximum
                    mx[t][k] = fmaxf(mx[t][k], av[r][t][k]);
                }
        bool big = false;
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) big = big || mx[t][k] > RESCALE_THR;
        if (hw::any(big)) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d = mx[t][k];
                    m[t][k] += d;
                    l[t][k] *= hw::exp2(-d);
#pragma unroll
                    for (int r = 0; r < 4; ++r) av[r][t][k] -= d;
                }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k) l[t][k] += hw::exp2(av[r][t][k]);
        hw::keep_alive(pk);
        if (st) t_i += hw::clock() - tq;
    };
    const int nqf = N >> 2;                          // quads that lie entirely below N
    for (int q = 0; q < nqf; ++q) stats_quad(std::false_type{}, q);
    if (nqf < nq) stats_quad(std::true_type{}, nqf);
    const unsigned long long t1 = st ? hw::clock() : 0, w1 = t_wait, s1 = t_s, i1 = t_i;
    f32x4 nlse[RT];                                  // minus the log2-domain log-sum-exp: initial accumulator of pass 2
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int g = 16 * t + 4 * g4 + k;
            const float lse2 = m[t][k] + hw::log2(l[t][k]);
            nlse[t][k] = -lse2;
            if (iok && g < H) p.stats[((long)b * H + g) * N + i] = lse2 * LN2;
        }

    // ---- pass 2: P = exp2(A' - lse'), R = conv_w(P), 8 keys per 16-byte store ------------------------------------------------------
    // R leaves as 8-byte groups of 4 keys.  The stores of quad u are issued AFTER the LDS-DMA request made at the top of quad u + 1:
    // vmcnt retires in issue order, so a store issued before a request would have to be acknowledged before that request's data
    // counts as landed.  Addresses: one 32-bit lane offset on top of wave-uniform bases.
    // quad-blocked layout [B, H, Np / 4, N, 4]: the 16 queries of the tile are contiguous (128 B) for a (head, quad)
    char* const rbase = (char*)(p.R + (long)b * H * N * p.Np);
    const unsigned rhead = (unsigned)(N * p.Np) * 2u;                       // bytes between heads
    const unsigned rquad = (unsigned)N * 8u;                                // bytes between quads
    const unsigned rlane = 4u * g4 * rhead + 8u * i;
    bf16x4 rp[RT][4];
    auto flush = [&](int u) {
        if (iok) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (16 * t + 4 * g4 + k < H) *(bf16x4*)(rbase + (unsigned)(16 * t + k) * rhead + (unsigned)u * rquad + rlane) = rp[t][k];
        }
    };
    auto quad2 = [&](auto tail_c, int u) {
        constexpr bool TAIL = decltype(tail_c)::value;
        const int j0 = 4 * u;
        if (st) tq = hw::clock();
        if (u + 1 < nq) ks.advance(u + 1); else hw::dma_wait<0>();
        if (st) { const unsigned long long t = hw::clock(); t_wait += t - tq; tq = t; }
        if (u > 0) flush(u - 1);
        f32x4 sa[NS];
        bd_scores<C>(ks.stage(0), zeros, lane, qf, sa);
        ks.done();
        if (st) { const unsigned long long t = hw::clock(); t_s += t - tq; tq = t; }
        // stage-wise over the quad's four keys: 4 x RT independent MFMAs, then 4 x RT x 4 exponentials, ... (the chains of the four
        // keys are independent: issued round-robin they hide each other's MFMA / transcendental latency)
        f32x4 a[4][RT];
        f16x8 pk[4], pp[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pk[r] = pack_s<C, f16x8>(sa, r);
#pragma unroll
            for (int t = 0; t < RT; ++t) a[r][t] = hw::mfma_f16(w.wl[t], pk[r], nlse[t]);
        }
        hw::sched_fence();
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k) a[r][t][k] = (!TAIL || j0 + r < N) ? hw::exp2(a[r][t][k]) : 0.f;
        hw::keep_alive(pk);
        hw::sched_fence();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pp[r] = pack_a<C, f16x8>(a[r]);
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                a[r][t] = hw::mfma_f16(w.ww[t], pp[r], z);
            }
        }
        hw::sched_fence();
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) rp[t][k] = bf16x4{(bf16_t)a[0][t][k], (bf16_t)a[1][t][k], (bf16_t)a[2][t][k], (bf16_t)a[3][t][k]};
        hw::keep_alive(pp);
        if (st) t_i += hw::clock() - tq;
    };
    for (int u = 0; u < nqf; ++u) quad2(std::false_type{}, u);
    if (nqf < nq) quad2(std::true_type{}, nqf);
    flush(nq - 1);
    if (nq & 1) {                                    // Np is a multiple of 8: the pad columns of a half-filled last block are zero
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) rp[t][k] = bf16x4{};
        flush(nq);
    }
    if (st && lane == 0) {
        unsigned long long* o = p.stamps + 12 * ((long)b * p.QT + it);
        const unsigned long long t2 = hw::clock();
        o[0] = t1 - t0; o[1] = w1; o[2] = s1; o[3] = i1;                          // pass 1: total, ring wait, scores, per-key stage
        o[4] = t2 - t1; o[5] = t_wait - w1; o[6] = t_s - s1; o[7] = t_i - i1;      // pass 2
    }
}

// the wave's zero block (read by the 48 lanes of a block-diagonal A operand that carry no row)
template <class C>
DEVFN void zero_block_init(char* lds, int lane) {
    for (int o = lane * 16; o < Ring<C>::ROWB; o += 64 * 16) *(u32x4*)(lds + o) = u32x4{0u, 0u, 0u, 0u};
    hw::lds_fence();
}

// ---- backward --------------------------------------------------------------------------------------------------------------
// weight-gradient products go through a wave-private LDS tile per operand: [32 head rows][64 elements] bf16, rows padded to 144 B
constexpr int WG_ROWB = 144;
constexpr int WG_TILE = 32 * WG_ROWB;            // 4608 B
template <class C>
constexpr int bwd_tile_off() { return Ring<C>::ROWB + 4 * Ring<C>::STAGE; }              // zero block, key ring, value ring
template <class C>
constexpr int bwd_lds_per_wave() { return bwd_tile_off<C>() + 2 * WG_TILE; }              // ... + the two tiles (9216 B >= 2 * 32 * 32 * 4)

template <class C>
struct BwdWeights {
    f16x8 wl[C::RT];      // forward mix 1 (as the forward: log2(e) * scale * W_l)
    bf16x8 wwt[C::RT];    // dP_h = sum_g W_w[g, h] dR_g   : out = h, in = g, first-kind layout (dR comes from block-diagonal products)
    bf16x8 wlt[C::RT];    // dS_h = sum_g W_l[g, h] dA_g   : out = h, in = g, second-kind layout
};

template <class C, bool PASS_B>
DEVFN void bwd_load_weights(const BwdArgs& p, int lane, BwdWeights<C>& w) {
    const float* Wl = p.Wl; const float* Ww = p.Ww;
    const float sc = p.scale * LOG2E;
#pragma unroll
    for (int t = 0; t < C::RT; ++t) {
        w.wl[t] = weight_frag<C, true, f16x8>(lane, t, [&](int g, int h) { return Wl[g * C::H + h] * sc; });
        w.wwt[t] = weight_frag<C, true, bf16x8>(lane, t, [&](int h, int g) { return Ww[g * C::H + h]; });
        if (PASS_B) w.wlt[t] = weight_frag<C, false, bf16x8>(lane, t, [&](int h, int g) { return Wl[g * C::H + h]; });
    }
}

// tile rows of a first-kind tensor (value v[s][r]: head 4s + g4, element (c, r)) / second-kind tensor (x[r][t][k]: head 16t + 4g4 + k)
template <class C>
DEVFN void wg_store_s(char* tile, int lane, const f32x4 (&v)[C::NS], float mul) {
    const int c = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int s = 0; s < C::NS; ++s) {
        const bf16x4 o = {(bf16_t)(v[s][0] * mul), (bf16_t)(v[s][1] * mul), (bf16_t)(v[s][2] * mul), (bf16_t)(v[s][3] * mul)};
        *(bf16x4*)(tile + (4 * s + g4) * WG_ROWB + c * 8) = o;
    }
}
template <class C>
DEVFN void wg_store_a(char* tile, int lane, const f32x4 (&x)[4][C::RT]) {
    const int c = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int t = 0; t < C::RT; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bf16x4 o = {(bf16_t)x[0][t][k], (bf16_t)x[1][t][k], (bf16_t)x[2][t][k], (bf16_t)x[3][t][k]};
            *(bf16x4*)(tile + (16 * t + 4 * g4 + k) * WG_ROWB + c * 8) = o;
        }
}
// acc[t][u] (rows g = 16t + .., columns h = 16u + ..) += sum over the 64 elements of X[g][e] Y[h][e]
template <class C>
DEVFN void wg_product(const char* tx, const char* ty, int lane, f32x4 (&acc)[C::RT][C::RT]) {
    const int c = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 xa[C::RT], yb[C::RT];
#pragma unroll
        for (int t = 0; t < C::RT; ++t) {
            xa[t] = *(const bf16x8*)(tx + (16 * t + c) * WG_ROWB + (32 * ks + 8 * g4) * 2);
            yb[t] = *(const bf16x8*)(ty + (16 * t + c) * WG_ROWB + (32 * ks + 8 * g4) * 2);
        }
#pragma unroll
        for (int t = 0; t < C::RT; ++t)
#pragma unroll
            for (int u = 0; u < C::RT; ++u) acc[t][u] = hw::mfma_bf16(xa[t], yb[u], acc[t][u]);
#pragma unroll
        for (int t = 0; t < C::RT; ++t) { hw::keep_alive(xa[t]); hw::keep_alive(yb[t]); }
    }
    hw::mfma_src_guard();                    // 2 x RT^2 MFMAs in a row: the last fragments must survive the queue
}

// One (sample, 16-query tile) of backward pass A (PASS_B = false): delta_h[i] = sum_j P_h dP_h -> p.delta, dW_w += dR P^T;
// or of pass B (PASS_B = true): dA = P o (dP - delta), dS = conv_l^T(dA) -> p.dS, dW_l += dA S^T.  Two launches instead of two
// passes of one kernel: each holds ONE weight-gradient accumulator and only its own mix operands next to the 2 x 96 fragment
// registers of q and dO (one kernel with both spilled), and delta [B, H, N] f32 is a 2.4 MB round trip.
template <class C, bool PASS_B>
DEVFN void bwd_item(const BwdArgs& p, int b, int it, int lane, const BwdWeights<C>& w, f32x4 (&acc)[C::RT][C::RT], char* lds) {
    constexpr int NS = C::NS, RT = C::RT, H = C::H, D = C::D;
    const int c = lane & 15, g4 = lane >> 4;
    const int N = p.N;
    const int i = it * 16 + c;
    const bool iok = i < N;
    char* tx = lds + bwd_tile_off<C>();
    char* ty = tx + WG_TILE;
    const char* zeros = lds;
    const bf16_t* base = p.qkv + (long)b * N * p.ld;
    bf16x8 qf[NS][C::NC], dof[NS][C::NC];
    load_col_frags<C, true>(base, p.ld, i, iok, lane, qf);
    load_col_frags<C, true>(p.dO + (long)b * N * p.ldo, p.ldo, i, iok, lane, dof);
    RowStream<C, 2> kv{{base + D, base + 2 * D}, p.ld, N, lds + Ring<C>::ROWB, lane, 0};
    f32x4 nlse[RT], delta[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int g = 16 * t + 4 * g4 + k;
            const bool ok = iok && g < H;
            // padded heads and queries beyond N: P = 0, so they drop out of every sum below
            nlse[t][k] = ok ? -p.stats[((long)b * H + g) * N + i] * LOG2E : -P_OFF;
            delta[t][k] = (PASS_B && ok) ? p.delta[((long)b * H + g) * N + i] : 0.f;
        }
    const int nq = (N + 3) >> 2, nqf = N >> 2;

    // dS leaves like R in the forward: 8-byte groups of 4 keys in the quad-blocked layout, issued after the next quad's requests
    char* const sbase = (char*)(p.dS + (long)b * H * N * p.Np);
    const unsigned shead = (unsigned)(N * p.Np) * 2u, squad = (unsigned)N * 8u;
    const unsigned slane = 4u * g4 * shead + 8u * i;
    bf16x4 sp[RT][4];
    auto flush = [&](int u) {
        if (iok) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (16 * t + 4 * g4 + k < H) *(bf16x4*)(sbase + (unsigned)(16 * t + k) * shead + (unsigned)u * squad + slane) = sp[t][k];
        }
    };

    const bool st = p.stamps != nullptr;
    unsigned long long t_wait = 0, t_s = 0, t_i = 0, t_w = 0, t0 = st ? hw::clock() : 0, tq = 0;
    kv.issue(0, 0);
    auto quad = [&](auto tail_c, int u) {
        constexpr bool TAIL = decltype(tail_c)::value;
        const int j0 = 4 * u;
        if (st) tq = hw::clock();
        if (u + 1 < nq) kv.advance(u + 1); else hw::dma_wait<0>();
        if (st) { const unsigned long long t = hw::clock(); t_wait += t - tq; tq = t; }
        if (PASS_B && u > 0) flush(u - 1);
        // the weight-gradient product of the PREVIOUS quad: its tiles were written at the end of that iteration, so neither the
        // LDS write latency nor the read latency sits on this iteration's critical path
        if (u > 0) wg_product<C>(tx, ty, lane, acc);
        f32x4 sa[NS], dra[NS], pr[4][RT], dp[4][RT];
        bd_scores<C>(kv.stage(0), zeros, lane, qf, sa);        // S  = q k^T (raw)
        bd_scores<C>(kv.stage(1), zeros, lane, dof, dra);      // dR = dO v^T
        kv.done();
        if (st) { const unsigned long long t = hw::clock(); t_s += t - tq; tq = t; }
        // stage-wise over the quad's four keys, as the forward
        f16x8 pk[4];
        bf16x8 dk[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pk[r] = pack_s<C, f16x8>(sa, r);
            dk[r] = pack_s<C, bf16x8>(dra, r);
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                pr[r][t] = hw::mfma_f16(w.wl[t], pk[r], nlse[t]);
                dp[r][t] = hw::mfma_bf16(w.wwt[t], dk[r], z);  // dP = conv_w^T(dR)
            }
        }
        hw::sched_fence();
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k) pr[r][t][k] = (!TAIL || j0 + r < N) ? hw::exp2(pr[r][t][k]) : 0.f;     // keys beyond N: P = 0
        if (!PASS_B) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int k = 0; k < 4; ++k) delta[t][k] += pr[r][t][k] * dp[r][t][k];
            if (st) { const unsigned long long t = hw::clock(); t_i += t - tq; tq = t; }
            // dW_w[g, h] += sum_e dR_g[e] P_h[e]: operands to the tiles now, product at the top of the next iteration
            hw::lds_fence();
            wg_store_s<C>(tx, lane, dra, 1.f);
            wg_store_a<C>(ty, lane, pr);
            hw::lds_fence();
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int k = 0; k < 4; ++k) dp[r][t][k] = pr[r][t][k] * (dp[r][t][k] - delta[t][k]);     // dA
            hw::sched_fence();
            // dS = conv_l^T(dA), two keys at a time (their packed pair is half of the 8-byte store)
            bf16x8 da[4];
#pragma unroll
            for (int r2 = 0; r2 < 4; r2 += 2) {
                f32x4 ds[2][RT];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    da[r2 + e] = pack_a<C, bf16x8>(dp[r2 + e]);
#pragma unroll
                    for (int t = 0; t < RT; ++t) {
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        ds[e][t] = hw::mfma_bf16(w.wlt[t], da[r2 + e], z);
                    }
                }
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int k = 0; k < 4; ++k) { sp[t][k][r2] = (bf16_t)ds[0][t][k]; sp[t][k][r2 + 1] = (bf16_t)ds[1][t][k]; }
            }
            if (st) { const unsigned long long t = hw::clock(); t_i += t - tq; tq = t; }
            // dW_l[g, h] += sum_e dA_g[e] S_h[e]   (S = scale * raw scores): product at the top of the next iteration
            hw::lds_fence();
            wg_store_a<C>(tx, lane, dp);
            wg_store_s<C>(ty, lane, sa, p.scale);
            hw::lds_fence();
            hw::keep_alive(da);                      // the four mix MFMAs above queue up: their operands stay put until here
        }
        // MFMA operands built by the VALU outlive their (queued) MFMAs by at least a dozen issue slots: hw::keep_alive
        hw::keep_alive(pk);
        hw::keep_alive(dk);
        if (st) t_w += hw::clock() - tq;
    };
    for (int u = 0; u < nqf; ++u) quad(std::false_type{}, u);
    if (nqf < nq) quad(std::true_type{}, nqf);
    wg_product<C>(tx, ty, lane, acc);                  // the last quad's
    if (PASS_B) {
        flush(nq - 1);
        if (nq & 1) {                                // Np is a multiple of 8: the pad columns of a half-filled last block are zero
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k) sp[t][k] = bf16x4{};
            flush(nq);
        }
    } else {
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int g = 16 * t + 4 * g4 + k;
                if (iok && g < H) p.delta[((long)b * H + g) * N + i] = delta[t][k];
            }
    }
    if (st && lane == 0) {
        unsigned long long* o = p.stamps + 8 * ((long)b * p.QT + it);
        o[0] = hw::clock() - t0; o[1] = t_wait; o[2] = t_s; o[3] = t_i; o[4] = t_w;     // total, ring wait, S + dR, per-key stage, dW product
    }
}

// Persistent wave: items wave0, wave0 + nwaves, ...; at the end the workgroup's waves add their weight-gradient tiles through
// LDS and the workgroup writes ONE partial [HP][HP] into slot `PASS_B ? 0 : 1` of its [2][HP][HP] record (summed by a tiny
// launch afterwards: no same-line atomics, run-to-run identical).
template <class C, bool PASS_B>
DEVFN void bwd_wave(const BwdArgs& p, int wg, int nwg, int wave, int nwave, int lane, char* lds_all) {
    constexpr int RT = C::RT, HP = C::HP;
    constexpr int PER_WAVE = bwd_lds_per_wave<C>(), TILE_OFF = bwd_tile_off<C>();
    char* lds = lds_all + wave * PER_WAVE;
    // the zero block; rows of the tiles that no product writes stay finite
    for (int o = lane * 16; o < 2 * WG_TILE; o += 64 * 16) *(u32x4*)(lds + TILE_OFF + o) = u32x4{0u, 0u, 0u, 0u};
    zero_block_init<C>(lds, lane);
    BwdWeights<C> w;
    bwd_load_weights<C, PASS_B>(p, lane, w);
    f32x4 acc[RT][RT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int u = 0; u < RT; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nitem = p.B * p.QT;
    for (int item = wg * nwave + wave; item < nitem; item += nwg * nwave) {
        const int b = item / p.QT, it = item - b * p.QT;
        bwd_item<C, PASS_B>(p, b, it, lane, w, acc, lds);
    }
    // accumulator layout: row g = 16t + 4 g4 + r, column h = 16u + c
    hw::lds_fence();
    float* mine = (float*)(lds + TILE_OFF);          // [HP][HP] f32
    const int c = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int u = 0; u < RT; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(16 * t + 4 * g4 + r) * HP + 16 * u + c] = acc[t][u][r];
    hw::block_sync();
    float* out = p.partial + ((long)wg * 2 + (PASS_B ? 0 : 1)) * HP * HP;
    for (int idx = wave * 64 + lane; idx < HP * HP; idx += nwave * 64) {
        float s = 0.f;
        for (int v = 0; v < nwave; ++v) s += ((const float*)(lds_all + v * PER_WAVE + TILE_OFF))[idx];
        out[idx] = s;
    }
}

}  // namespace amix
