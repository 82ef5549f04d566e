// Tower-level runtime: one C call = the whole stream-ordered launch sequence of an encoder tower.
//
//   CLIP tower (kind 0 frozen teacher / kind 2 trainable: the plain ImageEncoder / TextEncoder in the student role, reference
//                           image_encoder.py:23-25,54-59, text_encoder.py:45-47,75-80):
//                           reference model/component/_common.py:188-221 (VisionTransformer.forward),
//                           model/component/text_encoder.py:62-92 (TextEncoder.encode_text)
//   student (weight-shared MiniViT blocks): reference model/component/weight_share_model.py:336-372, :482-512
//                           (forward_features), :199-218 (RepeatedMiniBlock), :179-185 (MiniBlock), :88-140 (MiniAttention)
//
// The handle is an immutable plan (shapes + workspace/weight-cache offsets).  All device memory is owned by the caller:
//   params / grads : arrays of f32 device pointers in the canonical order documented in include/dclip.h
//   wcache         : bf16 copies of the GEMM weights (W and, for the student, W^T for dgrad), refreshed by _prepare
//   workspace      : activations; in training mode everything backward needs stays resident between the two calls
// No allocation, no synchronisation, no global state: safe to call from the autograd thread and capturable in a hipGraph.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <new>
#include <vector>
#include "common.h"

namespace {

#define CK(expr)                      \
    do {                              \
        int _rc = (expr);             \
        if (_rc != DCLIP_OK) return _rc; \
    } while (0)

inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Bump {
    char* base; size_t off;
    explicit Bump(void* b) : base((char*)b), off(0) {}
    template <class T> T* take(size_t n) {
        T* p = base ? (T*)(base + off) : nullptr;
        off += up256(n * sizeof(T));
        return p;
    }
};

// canonical parameter order (see include/dclip.h)
enum { P_PER_TBLOCK = 12, P_PER_SBLOCK = 8, P_PER_SREPEAT = 6 };

struct Plan {
    dclip_encoder_cfg c;
    int D, H, hd, N, Np, F, E, L, R, K;        // K = patch GEMM contraction (image)
    bool student, image, compressed;           // student = the weight-shared MiniViT architecture (kind 1)
    bool train;                                // the tower has a backward (kind 1, 2): transposed weights cached, f32 residual stream
    // parameter indices
    int p_embed0;                              // first embedding parameter
    int p_blocks;                              // first block parameter
    int p_final;                               // norm / ln_post .. projection
    int n_params;
    // weight cache offsets (bf16 elements), -1 = absent
    struct BlockW { int64_t qkv, qkv_t, proj, proj_t, fc1, fc1_t, fc2, fc2_t; };
    std::vector<BlockW> bw;
    int64_t w_embed, w_embed_t, w_head, w_head_t, w_total;
};

int64_t wtake(int64_t& off, int64_t n) { int64_t o = off; off += (n + 127) & ~(int64_t)127; return o; }

bool make_plan(const dclip_encoder_cfg& c, Plan& p) {
    p.c = c;
    p.student = c.kind == 1; p.train = c.kind != 0; p.image = c.modality == 0;
    p.D = c.width; p.H = c.heads; p.N = c.tokens; p.F = c.mlp_dim; p.E = c.out_dim; p.L = c.layers; p.R = c.repeats;
    if (c.kind < 0 || c.kind > 2 || c.modality < 0 || c.modality > 1) { dclip_set_error("encoder: bad kind/modality"); return false; }
    if (p.D <= 0 || p.H <= 0 || p.D % p.H) { dclip_set_error("encoder: width %d not divisible by heads %d", p.D, p.H); return false; }
    p.hd = p.D / p.H;
    if (p.hd != 32 && p.hd != 64) { dclip_set_error("encoder: head dim %d unsupported (32 or 64)", p.hd); return false; }
    if (p.D % 64 || p.F % 64 || p.E % 64 || p.D > 1024) { dclip_set_error("encoder: width/mlp/out dims must be multiples of 64, width <= 1024"); return false; }
    if (p.N <= 0 || p.N > 128) { dclip_set_error("encoder: tokens must be in 1..128 (got %d)", p.N); return false; }
    if (p.L <= 0 || p.R <= 0 || (!p.student && p.R != 1)) { dclip_set_error("encoder: bad layers/repeats"); return false; }
    p.Np = (p.N + 7) & ~7;
    p.compressed = !p.image && c.embed_rank > 0;
    p.K = 0;
    if (p.image) {
        if (c.patch <= 0 || c.resolution < c.patch || c.in_chans <= 0) { dclip_set_error("encoder: bad patch geometry"); return false; }
        const int g = c.resolution / c.patch;
        if (g * g + 1 != p.N) { dclip_set_error("encoder: tokens %d != (res/patch)^2 + 1 = %d", p.N, g * g + 1); return false; }
        p.K = c.in_chans * c.patch * c.patch;
        if (p.K % 64) { dclip_set_error("encoder: in_chans*patch^2 = %d must be a multiple of 64", p.K); return false; }
    } else {
        if (c.vocab <= 0) { dclip_set_error("encoder: vocab required for text"); return false; }
        if (p.compressed && c.embed_rank % 64) { dclip_set_error("encoder: embed_rank must be a multiple of 64"); return false; }
        if (p.compressed && !p.student) { dclip_set_error("encoder: a compressed token embedding exists for the weight-shared student only (kind 1)"); return false; }
    }
    if (p.student && c.head_mix && p.H != 2 && p.H != 4 && p.H != 8 && p.H != 12 && p.H != 24) {
        dclip_set_error("encoder: head count %d unsupported by the head-mixing kernels", p.H); return false;
    }
    // parameter order
    int n = 0;
    p.p_embed0 = 0;
    if (p.student) n += p.image ? 4 : (p.compressed ? 4 : 2);     // see header
    else n += p.image ? 5 : 2;
    p.p_blocks = n;
    n += p.student ? p.L * (P_PER_SBLOCK + p.R * P_PER_SREPEAT) : p.L * P_PER_TBLOCK;
    p.p_final = n;
    n += p.student ? 4 : 3;
    p.n_params = n;
    // weight cache
    int64_t off = 0;
    const int64_t D = p.D, F = p.F, E = p.E;
    p.bw.resize(p.L);
    for (int l = 0; l < p.L; ++l) {
        auto& b = p.bw[l];
        b.qkv = wtake(off, 3 * D * D); b.proj = wtake(off, D * D); b.fc1 = wtake(off, F * D); b.fc2 = wtake(off, D * F);
        if (p.train) { b.qkv_t = wtake(off, 3 * D * D); b.proj_t = wtake(off, D * D); b.fc1_t = wtake(off, F * D); b.fc2_t = wtake(off, D * F); }
        else b.qkv_t = b.proj_t = b.fc1_t = b.fc2_t = -1;
    }
    p.w_embed = p.w_embed_t = -1;
    if (p.image) p.w_embed = wtake(off, D * p.K);
    else if (p.compressed) { p.w_embed = wtake(off, D * c.embed_rank); p.w_embed_t = wtake(off, D * c.embed_rank); }
    p.w_head = wtake(off, E * D);                                 // [E, D] (CLIP towers: proj^T)
    p.w_head_t = p.train ? wtake(off, E * D) : -1;                // [D, E]
    p.w_total = off;
    return true;
}

// ------------------------------------------------------------------------------------------------------------
// workspace layout
// ------------------------------------------------------------------------------------------------------------
struct ExecSave {          // per block execution (student training)
    void* x_mid;           // residual stream after the attention branch: f32 (students), f16 (the frozen teacher)
    float* mean1; float* rstd1; float* mean2; float* rstd2;
    bf16_t *h1, *qkv, *P, *Rm, *ctx, *h2, *u;
    uint8_t* z;            // gelu'(fc1 pre-activation) as 8-bit fixed point (DCLIP_ACT_GELU_SAVE -> DCLIP_ACT_MULAUX)
    float* S;
    float* stats;          // [B, H, N] log-sum-exp rows of the register-resident attention path
};

struct Work {
    // persistent
    std::vector<void*> X;                  // residual stream: X[0] embedding output .. X[LR]; f32 for students, f16 for the frozen
                                           // teacher (what the reference's `precision: 16` autocast keeps there: _common.py:14-20, :124-125)
    bool h16;
    std::vector<ExecSave> ex;
    bf16_t* patches;                       // image: [M, K] ; compressed text: [M, rank]
    float* tok_table;                      // [N, D]
    int32_t* pick;                         // [B]
    float *meanf, *rstdf;                  // final LN stats [B]
    bf16_t* hf;                            // [B, D]
    // temporaries
    void* x0;                              // CLIP image tower: pre-ln_pre tokens (f16 frozen, f32 trainable)
    float *mean0, *rstd0;                  // ln_pre statistics (trainable CLIP image tower)
    float* G0;                             // gradient of the pre-ln_pre tokens (trainable CLIP image tower)
    float* G; bf16_t* Gb;                  // residual-stream gradient
    // gradients that are wgrad operands keep one slot per repeat: the R executions of a weight-shared block feed ONE
    // wgrad GEMM over R * M rows (half the launches and half the f32 atomic traffic at R = 2)
    bf16_t *gb_f2, *gb_pr;                 // [R][M, D] bf16 residual gradient as seen by fc2 / attn.proj
    bf16_t *dbig, *dh, *dqkv, *dR, *dS, *dout;   // dbig [R][M, F], dqkv [R][M, 3D]
    float* tok_sum; float* demb;           // [N, D] ; compressed: [M, rank] f32
    float* wg_dummy;                       // [2, H, H] sink for conv_l / conv_w gradients when those parameters are frozen
    float* mix_ws; size_t mix_ws_bytes;    // per-workgroup weight-gradient partials of dclip_attn_mix_bwd
    void* tn_ws; size_t tn_ws_bytes;       // partial tiles of the 256 x 256 wgrad launches (dclip_gemm_tn_acc)
    size_t bytes;
};

void layout(const Plan& p, int64_t B, bool training, void* base, Work& w, int64_t N = 0) {
    Bump b(base);
    if (N <= 0) N = p.N;
    const int64_t Npad = (N + 7) & ~(int64_t)7;
    const int64_t M = B * N, D = p.D, F = p.F, SN = B * p.H * N * Npad;
    const int nex = p.L * p.R;
    const bool save = p.train && training;
    w.X.assign(nex + 1, nullptr);
    w.ex.assign(nex, ExecSave{});
    // teacher / inference: a single ping-pong set reused by every block
    ExecSave shared{};
    void* xs = nullptr;
    w.h16 = !p.train;
    if (!save) {
        xs = w.h16 ? (void*)b.take<_Float16>(M * D) : (void*)b.take<float>(M * D);
        shared.x_mid = xs;   // in-place residual stream
        shared.h1 = b.take<bf16_t>(M * D); shared.qkv = b.take<bf16_t>(M * 3 * D);
        shared.S = b.take<float>(SN); shared.P = (p.student && p.c.head_mix) ? b.take<bf16_t>(SN) : nullptr;
        shared.Rm = b.take<bf16_t>(SN);
        shared.stats = b.take<float>(B * p.H * N);
        shared.ctx = b.take<bf16_t>(M * D); shared.h2 = shared.h1; shared.z = nullptr; shared.u = b.take<bf16_t>(M * F);
        shared.mean1 = shared.rstd1 = shared.mean2 = shared.rstd2 = nullptr;
    }
    for (int e = 0; e <= nex; ++e) w.X[e] = save ? (void*)b.take<float>(M * D) : xs;
    for (int e = 0; e < nex; ++e) {
        if (!save) { w.ex[e] = shared; continue; }
        ExecSave& s = w.ex[e];
        s.x_mid = b.take<float>(M * D);
        s.mean1 = b.take<float>(M); s.rstd1 = b.take<float>(M); s.mean2 = b.take<float>(M); s.rstd2 = b.take<float>(M);
        s.qkv = b.take<bf16_t>(M * 3 * D);
        s.S = b.take<float>(SN); s.P = b.take<bf16_t>(SN); s.Rm = p.c.head_mix ? b.take<bf16_t>(SN) : s.P;
        s.stats = b.take<float>(B * p.H * N);
        s.z = b.take<uint8_t>(M * F);
        if (e % p.R == 0) {
            // the wgrad operands (inputs of the four linears) of a block's R executions lie back to back: [R][M, .]
            bf16_t* h1 = b.take<bf16_t>(p.R * M * D); bf16_t* ctx = b.take<bf16_t>(p.R * M * D);
            bf16_t* h2 = b.take<bf16_t>(p.R * M * D); bf16_t* u = b.take<bf16_t>(p.R * M * F);
            for (int r = 0; r < p.R; ++r) {
                ExecSave& t = w.ex[e + r];
                t.h1 = h1 + r * M * D; t.ctx = ctx + r * M * D; t.h2 = h2 + r * M * D; t.u = u + r * M * F;
            }
        }
    }
    w.patches = p.image ? b.take<bf16_t>(M * p.K) : (p.compressed ? b.take<bf16_t>(M * p.c.embed_rank) : nullptr);
    w.tok_table = b.take<float>((int64_t)N * D);
    w.pick = b.take<int32_t>(B);
    w.meanf = b.take<float>(B); w.rstdf = b.take<float>(B);
    w.hf = b.take<bf16_t>(B * D);
    w.x0 = (!p.student && p.image) ? (w.h16 ? (void*)b.take<_Float16>(M * D) : (void*)b.take<float>(M * D)) : nullptr;
    const bool pre = save && !p.student && p.image;
    w.mean0 = pre ? b.take<float>(M) : nullptr; w.rstd0 = pre ? b.take<float>(M) : nullptr; w.G0 = pre ? b.take<float>(M * D) : nullptr;
    if (save) {
        w.G = b.take<float>(M * D); w.Gb = b.take<bf16_t>(M * D);
        w.gb_f2 = b.take<bf16_t>(p.R * M * D); w.gb_pr = b.take<bf16_t>(p.R * M * D);
        w.dbig = b.take<bf16_t>(p.R * M * F); w.dh = b.take<bf16_t>(M * D); w.dqkv = b.take<bf16_t>(p.R * M * 3 * D);
        w.dR = b.take<bf16_t>(SN); w.dS = b.take<bf16_t>(SN); w.dout = b.take<bf16_t>(B * p.E);
        w.tok_sum = b.take<float>((int64_t)N * D);
        w.wg_dummy = b.take<float>(2 * p.H * p.H);
        w.mix_ws_bytes = p.c.head_mix ? dclip_attn_mix_bwd_workspace_bytes(B, p.H, N) : 0;
        w.mix_ws = w.mix_ws_bytes ? (float*)b.take<char>(w.mix_ws_bytes) : nullptr;
        w.tn_ws_bytes = dclip_gemm_tn_workspace_bytes();
        w.tn_ws = b.take<char>(w.tn_ws_bytes);
        w.demb = p.compressed ? b.take<float>(M * p.c.embed_rank) : nullptr;
    } else {
        w.G = nullptr; w.Gb = nullptr; w.gb_f2 = w.gb_pr = nullptr; w.dbig = w.dh = w.dqkv = w.dR = w.dS = w.dout = nullptr; w.tok_sum = w.demb = nullptr; w.wg_dummy = nullptr;
        w.mix_ws = nullptr; w.mix_ws_bytes = 0; w.tn_ws = nullptr; w.tn_ws_bytes = 0;
    }
    w.bytes = b.off;
}

inline const float* PF(const void* const* params, int i) { return (const float*)params[i]; }

// Head-mixing students whose shape has an instantiation of the register-resident score stage (attention_mix.hip: both head mixes
// on the matrix pipe) keep S, A, P, dR out of HBM; forward and backward must agree (the backward recomputes from qkv + the forward's
// softmax statistics), hence one predicate for both.  DCLIP_ATTN_MIX=0 selects the unfused kernels (attn_nt -> softmax -> ...).
inline bool mix_attn(const Plan& p, int64_t N) {
    static const int mode = [] { const char* e = getenv("DCLIP_ATTN_MIX"); return e ? atoi(e) : 1; }();
    return mode != 0 && p.student && p.c.head_mix && !p.c.causal && dclip_attn_mix_supported(p.H, N, p.hd) != 0;
}

// split count of the wgrad contraction: minimise  rounds(tiles*s / 512 resident workgroups) * work per workgroup
//                                                   + atomic traffic (s * P*Q*4 bytes at ~1.3 TB/s, half hidden)
inline int wsplits(int64_t M, int64_t P, int64_t Q) {
    const double tiles = (double)((P + 127) / 128) * (double)((Q + 127) / 128);
    const int smax = (int)((M + 255) / 256);
    int best = 1;
    double best_t = 1e30;
    for (int s = 1; s <= smax && s <= 64; ++s) {
        const double blocks = tiles * s;
        const double rounds = ceil(blocks / 512.0);
        const double t_mm = rounds * (2.0 * 128 * 128 * ((double)M / s)) / 1.76e12;      // ~900 TF/s shared by 512 workgroups
        const double t_at = 0.5 * s * (double)P * Q * 4.0 / 1.3e12;
        const double t = t_mm + t_at + 2e-6;
        if (t < best_t) { best_t = t; best = s; }
    }
    return best;
}

inline int gemm(const void* A, int64_t lda, const void* Bw, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                const float* bias, int act, const void* aux_in, void* aux_out, const void* res, int64_t ldr, int out_dtype,
                int64_t row_group, const float* rowadd, void* st) {
    return dclip_gemm_nt(A, lda, Bw, ldb, C, ldc, M, N, K, 1.f, bias, act, aux_in, aux_out, res, ldr, out_dtype, row_group, rowadd, nullptr, st);
}

// LayerNorm over rows of the tower's residual stream (f32, or f16 for the frozen teacher)
inline int ln_stream(bool h16, const void* x, int64_t ldx, const int32_t* ridx, const float* g, const float* b, void* y, int64_t ldy, int out_dtype,
                     float* mean, float* rstd, int64_t M, int64_t D, void* st) {
    if (h16) return dclip_layernorm_fwd_f16(x, ldx, ridx, g, b, y, ldy, out_dtype, mean, rstd, M, D, 1e-5f, st);
    return dclip_layernorm_fwd((const float*)x, ldx, ridx, g, b, y, ldy, out_dtype, mean, rstd, M, D, 1e-5f, st);
}

// a [M, D] piece of the residual stream as the f32 tensor the caller asked for (hidden-state / embedding export)
inline int export_stream(bool h16, const void* src, float* dst, int64_t n, void* st) {
    if (h16) return dclip_cast_f16_f32(src, dst, n, st);
    if (hipMemcpyAsync(dst, src, (size_t)n * 4, hipMemcpyDeviceToDevice, (hipStream_t)st) != hipSuccess) {
        dclip_set_error("dclip_encoder_forward: hidden-state export failed");
        return DCLIP_ELAUNCH;
    }
    return DCLIP_OK;
}

}  // namespace

// seeded: the workspace whose backward seeds (residual-gradient accumulator, its bf16 copy, the last execution's fc2 operand slot)
// the most recent training forward of this handle left cleared — the only mutable word of the handle.  dclip_encoder_backward
// consumes it; a backward that does not find its workspace there (a second backward on one forward, a retry after a failed one,
// another workspace in between) clears the seeds itself, so the call is self-contained whatever the caller does.
struct dclip_encoder { Plan p; mutable std::atomic<void*> seeded{nullptr}; };

extern "C" dclip_encoder* dclip_encoder_create(const dclip_encoder_cfg* cfg) {
    if (!cfg) { dclip_set_error("dclip_encoder_create: null cfg"); return nullptr; }
    dclip_encoder* e = new (std::nothrow) dclip_encoder();
    if (!e) { dclip_set_error("dclip_encoder_create: out of host memory"); return nullptr; }
    if (!make_plan(*cfg, e->p)) { delete e; return nullptr; }
    return e;
}

extern "C" void dclip_encoder_destroy(dclip_encoder* e) { delete e; }
extern "C" int64_t dclip_encoder_num_params(const dclip_encoder* e) { return e ? e->p.n_params : -1; }
extern "C" size_t dclip_encoder_wcache_bytes(const dclip_encoder* e) { return e ? up256((size_t)e->p.w_total * 2) : 0; }

extern "C" size_t dclip_encoder_workspace_bytes(const dclip_encoder* e, int64_t B, int training) {
    if (!e || B <= 0) return 0;
    Work w;
    layout(e->p, B, training != 0, nullptr, w);
    return w.bytes;
}

// parameter index helpers -------------------------------------------------------------------------------------
namespace {
struct TB { int ln1w, ln1b, inw, inb, outw, outb, ln2w, ln2b, fcw, fcb, prw, prb; };
inline TB tblock(const Plan& p, int l) {
    const int b = p.p_blocks + l * P_PER_TBLOCK;
    return TB{b, b + 1, b + 2, b + 3, b + 4, b + 5, b + 6, b + 7, b + 8, b + 9, b + 10, b + 11};
}
struct SB { int qkvw, qkvb, prw, prb, f1w, f1b, f2w, f2b; };
inline SB sblock(const Plan& p, int l) {
    const int b = p.p_blocks + l * (P_PER_SBLOCK + p.R * P_PER_SREPEAT);
    return SB{b, b + 1, b + 2, b + 3, b + 4, b + 5, b + 6, b + 7};
}
struct SR { int n1w, n1b, n2w, n2b, cl, cw; };
// one block execution's parameters under the names the backward uses, for both architectures (CLIP: ln_1/2, in_proj, out_proj, c_fc, c_proj)
struct BX { int n1w, n1b, n2w, n2b, qkvw, qkvb, prw, prb, f1w, f1b, f2w, f2b, cl, cw; };
inline SR srepeat(const Plan& p, int l, int r) {
    const int b = p.p_blocks + l * (P_PER_SBLOCK + p.R * P_PER_SREPEAT) + P_PER_SBLOCK + r * P_PER_SREPEAT;
    return SR{b, b + 1, b + 2, b + 3, b + 4, b + 5};
}
inline BX bexec(const Plan& p, int l, int r) {
    if (p.student) {
        const SB s = sblock(p, l); const SR q = srepeat(p, l, r);
        return BX{q.n1w, q.n1b, q.n2w, q.n2b, s.qkvw, s.qkvb, s.prw, s.prb, s.f1w, s.f1b, s.f2w, s.f2b, q.cl, q.cw};
    }
    const TB t = tblock(p, l);
    return BX{t.ln1w, t.ln1b, t.ln2w, t.ln2b, t.inw, t.inb, t.outw, t.outb, t.fcw, t.fcb, t.prw, t.prb, -1, -1};
}
}  // namespace

extern "C" int dclip_encoder_prepare(const dclip_encoder* e, const void* const* params, void* wcache, void* st) {
    DCLIP_REQUIRE(e && params && wcache, "dclip_encoder_prepare: null argument");
    const Plan& p = e->p;
    bf16_t* W = (bf16_t*)wcache;
    const int64_t D = p.D, F = p.F, E = p.E;
    auto at = [&](int64_t off) -> void* { return off < 0 ? nullptr : (void*)(W + off); };
    // one multi-tensor launch for the whole tower (students refresh their cache every step: 14 / 9 launches became 1)
    std::vector<const float*> src; std::vector<void*> wb, wt; std::vector<int64_t> rr, cc;
    auto job = [&](const float* w, void* b, void* t, int64_t R, int64_t C) { src.push_back(w); wb.push_back(b); wt.push_back(t); rr.push_back(R); cc.push_back(C); };
    for (int l = 0; l < p.L; ++l) {
        const auto& b = p.bw[l];
        int iq, ip, i1, i2;
        if (p.student) { SB s = sblock(p, l); iq = s.qkvw; ip = s.prw; i1 = s.f1w; i2 = s.f2w; }
        else { TB t = tblock(p, l); iq = t.inw; ip = t.outw; i1 = t.fcw; i2 = t.prw; }
        job(PF(params, iq), at(b.qkv), at(b.qkv_t), 3 * D, D);
        job(PF(params, ip), at(b.proj), at(b.proj_t), D, D);
        job(PF(params, i1), at(b.fc1), at(b.fc1_t), F, D);
        job(PF(params, i2), at(b.fc2), at(b.fc2_t), D, F);
    }
    if (p.image) job(PF(params, 0), at(p.w_embed), nullptr, D, p.K);                          // conv weight [D, C*p*p]
    else if (p.compressed) job(PF(params, 1), at(p.w_embed), at(p.w_embed_t), D, p.c.embed_rank);
    if (p.student) job(PF(params, p.p_final + 2), at(p.w_head), at(p.w_head_t), E, D);
    else job(PF(params, p.p_final + 2), at(p.w_head_t), at(p.w_head), D, E);                  // proj [D,E] -> [E,D] (+ as it is: the dgrad operand of a trainable tower)
    for (size_t i = 0; i < src.size(); ++i) DCLIP_REQUIRE(src[i], "dclip_encoder_prepare: parameter %zu missing", i);
    return dclip_cast_transpose_bf16_multi(src.data(), wb.data(), wt.data(), rr.data(), cc.data(), (int64_t)src.size(), st);
}

// The backward starts from a residual-stream gradient that is non-zero in B of the M rows (the picked class / EOT rows): its f32
// accumulator and the fc2 operand slot of the last execution start as zeros.  Those fills used to open dclip_encoder_backward, i.e. sat
// on the critical path right after the loss; issued at the end of the training forward, they run on the tower's stream while the other
// towers finish and the loss is evaluated (the buffers are not touched by the forward).  (Round 5: the bf16 copy w.Gb is no longer
// cleared — every row of it is written by the first execution's LayerNorm backward before anything reads it.)
static int clear_backward_seeds(const Plan& p, const Work& w, int64_t M, void* st) {
    const int64_t D = p.D;
    hipStream_t hs = (hipStream_t)st;
    bf16_t* gb_last = w.gb_f2 + (int64_t)(p.R - 1) * M * D;
    if (hipMemsetAsync(w.G, 0, (size_t)M * D * 4, hs) != hipSuccess || hipMemsetAsync(gb_last, 0, (size_t)M * D * 2, hs) != hipSuccess) {
        dclip_set_error("dclip_encoder: clearing the backward seeds failed");
        return DCLIP_ELAUNCH;
    }
    return DCLIP_OK;
}

// ext_patches: the image tower's [B*N, K] bf16 patch rows made by the caller (dclip_im2row, cls_rows = 1) — two towers that
// see the same images and cut them the same way share one conversion; null: this call converts `input` itself
static int encoder_forward_impl(const dclip_encoder* e, const void* input, const bf16_t* ext_patches, int64_t B, const void* const* params,
                                const void* wcache, void* workspace, size_t ws_bytes, int training,
                                float* last_representation, float* const* rep_out, float* emb_out, int64_t tokens_eff,
                                void* st) {
    DCLIP_REQUIRE(e && (input || ext_patches) && params && wcache && workspace && last_representation, "dclip_encoder_forward: null argument");
    DCLIP_REQUIRE(B > 0, "dclip_encoder_forward: empty batch");
    const Plan& p = e->p;
    DCLIP_REQUIRE(!ext_patches || (p.image && ((uintptr_t)ext_patches % 16) == 0), "dclip_encoder_forward_patches: image towers only, 16-byte aligned rows");
    DCLIP_REQUIRE(!training || p.train, "dclip_encoder_forward: the frozen teacher tower (kind 0) is inference-only");
    // tokens_eff: causal text teacher only.  Positions after the longest caption's EOT cannot influence any EOT row (causal
    // attention; LN / MLP are per token), so the tower may run on the first tokens_eff positions with identical output.
    DCLIP_REQUIRE(tokens_eff == 0 || (!p.train && !p.image && p.c.causal && tokens_eff > 0 && tokens_eff <= p.N && !rep_out && !emb_out),
                  "dclip_encoder_forward: tokens_eff is only valid for the causal text teacher without hidden-state export");
    Work w;
    layout(p, B, training != 0, workspace, w, tokens_eff);
    DCLIP_REQUIRE(ws_bytes >= w.bytes, "dclip_encoder_forward: workspace too small (%zu < %zu)", ws_bytes, w.bytes);
    DCLIP_REQUIRE(((uintptr_t)workspace % 256) == 0 && ((uintptr_t)wcache % 256) == 0, "dclip_encoder_forward: buffers must be 256-byte aligned");
    const bf16_t* W = (const bf16_t*)wcache;
    const int64_t N = tokens_eff ? tokens_eff : p.N, D = p.D, F = p.F, E = p.E, M = B * N, H = p.H, hd = p.hd, Np = (N + 7) & ~(int64_t)7;
    const float scale = 1.f / sqrtf((float)hd);
    const int nex = p.L * p.R;

    // ---- embedding -----------------------------------------------------------------------------------------
    if (p.image) {
        const bf16_t* patches = ext_patches ? ext_patches : w.patches;
        if (!ext_patches) CK(dclip_im2row((const float*)input, w.patches, B, p.c.in_chans, p.c.resolution, p.c.patch, 1, st));
        if (p.student) {   // params: 0 conv w, 1 conv b, 2 cls_token, 3 pos_embed
            CK(dclip_token_table(PF(params, 3), PF(params, 2), PF(params, 1), w.tok_table, N, D, st));
            CK(gemm(patches, p.K, W + p.w_embed, p.K, w.X[0], D, M, D, p.K, nullptr, 0, nullptr, nullptr, nullptr, 0, 1, N, w.tok_table, st));
        } else {           // params: 0 conv1 w, 1 class_embedding, 2 positional_embedding, 3 ln_pre w, 4 ln_pre b
            CK(dclip_token_table(PF(params, 2), PF(params, 1), nullptr, w.tok_table, N, D, st));
            const int sdt0 = w.h16 ? DCLIP_OUT_F16 : DCLIP_OUT_F32;
            CK(gemm(patches, p.K, W + p.w_embed, p.K, w.x0, D, M, D, p.K, nullptr, 0, nullptr, nullptr, nullptr, 0, sdt0, N, w.tok_table, st));
            CK(ln_stream(w.h16, w.x0, D, nullptr, PF(params, 3), PF(params, 4), w.X[0], D, sdt0, w.mean0, w.rstd0, M, D, st));
        }
    } else if (p.compressed) {   // params: 0 table [V,rank], 1 linear w [D,rank], 2 linear b, 3 pos
        CK(dclip_embed_gather((const int64_t*)input, p.N, PF(params, 0), nullptr, w.patches, 0, M, N, p.c.embed_rank, st));
        CK(dclip_token_table(PF(params, 3), nullptr, PF(params, 2), w.tok_table, N, D, st));
        CK(gemm(w.patches, p.c.embed_rank, W + p.w_embed, p.c.embed_rank, w.X[0], D, M, D, p.c.embed_rank, nullptr, 0, nullptr, nullptr, nullptr, 0, 1, N, w.tok_table, st));
    } else {                     // params: 0 table [V,D], 1 pos [N,D]
        CK(dclip_embed_gather((const int64_t*)input, p.N, PF(params, 0), PF(params, 1), w.X[0], w.h16 ? DCLIP_OUT_F16 : DCLIP_OUT_F32, M, N, D, st));
    }

    // optional export of the post-positional-embedding tokens (reference ControlOutput.need_emb: _common.py:204-206 captures
    // them BEFORE ln_pre; text_encoder.py:66-67 ; weight_share_model.py:350,490)
    if (emb_out) CK(export_stream(w.h16, (!p.student && p.image) ? w.x0 : w.X[0], emb_out, M * D, st));

    // ---- blocks --------------------------------------------------------------------------------------------
    for (int ei = 0; ei < nex; ++ei) {
        const int l = ei / p.R, r = ei % p.R;
        const auto& bw = p.bw[l];
        ExecSave& s = w.ex[ei];
        const float *n1w, *n1b, *n2w, *n2b, *bq, *bp, *b1, *b2, *wl = nullptr, *ww = nullptr;
        {
            const BX bx = bexec(p, l, r);
            n1w = PF(params, bx.n1w); n1b = PF(params, bx.n1b); n2w = PF(params, bx.n2w); n2b = PF(params, bx.n2b);
            bq = PF(params, bx.qkvb); bp = PF(params, bx.prb); b1 = PF(params, bx.f1b); b2 = PF(params, bx.f2b);
            if (p.student && p.c.head_mix) { wl = PF(params, bx.cl); ww = PF(params, bx.cw); }
        }
        void* xin = w.X[ei];
        void* xout = w.X[ei + 1];
        const int sdt = w.h16 ? DCLIP_OUT_F16 : DCLIP_OUT_F32;        // dtype of the residual stream
        CK(ln_stream(w.h16, xin, D, nullptr, n1w, n1b, s.h1, D, DCLIP_OUT_BF16, s.mean1, s.rstd1, M, D, st));
        CK(gemm(s.h1, D, W + bw.qkv, D, s.qkv, 3 * D, M, 3 * D, D, bq, 0, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, st));
        if (!training && !wl) {
            // inference without head mixing (the frozen teacher): one fused kernel, no score tensors in HBM
            CK(dclip_attn_fused_fwd(s.qkv, 3 * D, s.ctx, D, B, H, N, hd, scale, p.c.causal, st));
        } else {
            if (wl && mix_attn(p, N)) {
                CK(dclip_attn_mix_fwd(s.qkv, 3 * D, wl, ww, s.Rm, s.stats, B, H, N, Np, hd, scale, st));
                CK(dclip_attn_nn(s.Rm, s.qkv + 2 * D, 3 * D, s.ctx, D, B, H, N, Np, hd, 1.f, 1, st));
            } else {
                CK(dclip_attn_nt(s.qkv, 3 * D, s.qkv + D, 3 * D, s.S, 1, B, H, N, Np, hd, scale, st));
                CK(dclip_attn_softmax_fwd(s.S, wl, ww, wl ? s.P : nullptr, s.Rm, B, H, N, Np, p.c.causal, st));
                CK(dclip_attn_nn(s.Rm, s.qkv + 2 * D, 3 * D, s.ctx, D, B, H, N, Np, hd, 1.f, 0, st));
            }
        }
        CK(gemm(s.ctx, D, W + bw.proj, D, s.x_mid, D, M, D, D, bp, 0, nullptr, nullptr, xin, D, sdt, 0, nullptr, st));
        CK(ln_stream(w.h16, s.x_mid, D, nullptr, n2w, n2b, s.h2, D, DCLIP_OUT_BF16, s.mean2, s.rstd2, M, D, st));
        CK(gemm(s.h2, D, W + bw.fc1, D, s.u, F, M, F, D, b1, p.student ? (s.z ? DCLIP_ACT_GELU_SAVE : DCLIP_ACT_GELU) : (s.z ? DCLIP_ACT_QUICKGELU_SAVE : DCLIP_ACT_QUICKGELU), nullptr, s.z, nullptr, 0, 0, 0, nullptr, st));
        CK(gemm(s.u, F, W + bw.fc2, F, xout, D, M, D, F, b2, 0, nullptr, nullptr, s.x_mid, D, sdt, 0, nullptr, st));
        // optional export of this execution's hidden state (ControlOutput.need_rep: _common.py:156-158, weight_share_model.py:211)
        if (rep_out && rep_out[ei]) CK(export_stream(w.h16, xout, rep_out[ei], M * D, st));
    }

    // ---- final norm + projection on the picked token only (class token / EOT = argmax of the ids) ----------------
    CK(dclip_pick_index(p.image ? nullptr : (const int64_t*)input, p.N, w.pick, B, N, st));
    const int f = p.p_final;
    CK(ln_stream(w.h16, w.X[nex], D, w.pick, PF(params, f), PF(params, f + 1), w.hf, D, DCLIP_OUT_BF16, w.meanf, w.rstdf, B, D, st));
    CK(gemm(w.hf, D, W + p.w_head, D, last_representation, E, B, E, D, p.student ? PF(params, f + 3) : nullptr, 0, nullptr, nullptr, nullptr, 0, 1, 0, nullptr, st));
    if (training) {
        CK(clear_backward_seeds(p, w, M, st));
        e->seeded.store(workspace, std::memory_order_release);
    }
    return DCLIP_OK;
}

extern "C" int dclip_encoder_forward(const dclip_encoder* e, const void* input, int64_t B, const void* const* params,
                                     const void* wcache, void* workspace, size_t ws_bytes, int training,
                                     float* last_representation, float* const* rep_out, float* emb_out, int64_t tokens_eff,
                                     void* st) {
    DCLIP_REQUIRE(input, "dclip_encoder_forward: null argument");
    return encoder_forward_impl(e, input, nullptr, B, params, wcache, workspace, ws_bytes, training, last_representation, rep_out, emb_out,
                                tokens_eff, st);
}

extern "C" int dclip_encoder_forward_patches(const dclip_encoder* e, const void* patches, int64_t B, const void* const* params,
                                             const void* wcache, void* workspace, size_t ws_bytes, int training,
                                             float* last_representation, float* const* rep_out, float* emb_out, void* st) {
    DCLIP_REQUIRE(patches, "dclip_encoder_forward_patches: null argument");
    return encoder_forward_impl(e, nullptr, (const bf16_t*)patches, B, params, wcache, workspace, ws_bytes, training, last_representation,
                                rep_out, emb_out, 0, st);
}

// All-token output of the final norm + projection (reference _common.py:210-215, text_encoder.py:69-72,
// weight_share_model.py:363-366 / :503-506: `last_layer_output`, of which `last_representation` is one row per sample).  The
// training path projects only the picked row; this call produces the whole [B*N, E] tensor on request from the residual
// stream the most recent forward of this tower left in `workspace`.
extern "C" int dclip_encoder_last_layer_output(const dclip_encoder* e, int64_t B, const void* const* params, const void* wcache,
                                               void* workspace, size_t ws_bytes, int training, void* scratch, float* out, void* st) {
    DCLIP_REQUIRE(e && params && wcache && workspace && scratch && out, "dclip_encoder_last_layer_output: null argument");
    DCLIP_REQUIRE(B > 0, "dclip_encoder_last_layer_output: empty batch");
    const Plan& p = e->p;
    DCLIP_REQUIRE(!training || p.train, "dclip_encoder_last_layer_output: the frozen teacher tower (kind 0) is inference-only");
    Work w;
    layout(p, B, training != 0, workspace, w);
    DCLIP_REQUIRE(ws_bytes >= w.bytes, "dclip_encoder_last_layer_output: workspace too small (%zu < %zu)", ws_bytes, w.bytes);
    const bf16_t* W = (const bf16_t*)wcache;
    const int64_t M = B * p.N, D = p.D, E = p.E;
    const int nex = p.L * p.R, f = p.p_final;
    CK(ln_stream(w.h16, w.X[nex], D, nullptr, PF(params, f), PF(params, f + 1), scratch, D, DCLIP_OUT_BF16, nullptr, nullptr, M, D, st));
    CK(gemm(scratch, D, W + p.w_head, D, out, E, M, E, D, p.student ? PF(params, f + 3) : nullptr, 0, nullptr, nullptr, nullptr, 0, 1, 0, nullptr, st));
    return DCLIP_OK;
}

static int encoder_backward_impl(const dclip_encoder* e, const void* input, const bf16_t* ext_patches, int64_t B, const void* const* params,
                                 void* const* grads, const void* wcache, void* workspace, size_t ws_bytes,
                                 const float* d_last_representation, const float* const* d_rep, const float* d_emb,
                                 dclip_bucket_cb on_bucket, void* cb_user, void* st) {
    DCLIP_REQUIRE(e && (input || ext_patches) && params && grads && wcache && workspace && d_last_representation, "dclip_encoder_backward: null argument");
    const Plan& p = e->p;
    DCLIP_REQUIRE(!ext_patches || p.image, "dclip_encoder_backward_patches: image towers only");
    DCLIP_REQUIRE(p.train, "dclip_encoder_backward: the frozen teacher tower (kind 0) has no backward");
    Work w;
    layout(p, B, true, workspace, w);
    DCLIP_REQUIRE(ws_bytes >= w.bytes, "dclip_encoder_backward: workspace too small");
    const bf16_t* W = (const bf16_t*)wcache;
    const int64_t N = p.N, D = p.D, F = p.F, E = p.E, M = B * N, H = p.H, hd = p.hd, Np = p.Np;
    const float scale = 1.f / sqrtf((float)hd);
    const int nex = p.L * p.R;
    auto GR = [&](int i) -> float* { return (float*)grads[i]; };
    hipStream_t hs = (hipStream_t)st;

    // w.G, w.Gb and the last execution's fc2 operand slot were cleared at the end of the training forward of THIS workspace
    // (clear_backward_seeds) unless a backward has consumed them since: then they are cleared here
    {
        void* expect = workspace;
        if (!e->seeded.compare_exchange_strong(expect, nullptr, std::memory_order_acq_rel)) CK(clear_backward_seeds(p, w, M, st));
    }
    // ---- head + final norm -----------------------------------------------------------------------------------
    const int f = p.p_final;
    CK(dclip_cast_bf16(d_last_representation, w.dout, B * E, st));
    if (p.student) {         // head = nn.Linear: weight [E, D], bias
        if (GR(f + 2)) CK(dclip_gemm_tn_acc(w.dout, E, w.hf, D, GR(f + 2), D, B, E, D, 1, w.tn_ws, w.tn_ws_bytes, st));
        if (GR(f + 3)) CK(dclip_colsum_acc(w.dout, E, GR(f + 3), B, E, st));
    } else if (GR(f + 2)) {  // x @ proj: proj [D, E], no bias (reference _common.py:213, text_encoder.py:72)
        CK(dclip_gemm_tn_acc(w.hf, D, w.dout, E, GR(f + 2), E, B, D, E, 1, w.tn_ws, w.tn_ws_bytes, st));
    }
    CK(gemm(w.dout, E, W + p.w_head_t, E, w.dh, D, B, D, E, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, st));
    // every LayerNorm backward also emits the column sums of the updated residual gradient = the bias gradient of the
    // linear that wrote into that residual stream (fc2 of the previous execution / attn.proj of this one)
    const int R = p.R;
    bf16_t* gb_last = w.gb_f2 + (int64_t)(R - 1) * M * D;            // fc2 of the last execution reads slot R - 1
    CK(dclip_layernorm_bwd(w.dh, D, 0, (const float*)w.X[nex], D, w.pick, PF(params, f), w.meanf, w.rstdf, w.G, D, gb_last, D, GR(f), GR(f + 1),
                           GR(bexec(p, (nex - 1) / p.R, 0).f2b), B, D, st));
    // gradient bucket 0 (final norm + head) is complete: every launch that writes it is enqueued on `st`
    if (on_bucket) on_bucket(cb_user, 0);

    // ---- blocks, last execution first --------------------------------------------------------------------------
    for (int ei = nex - 1; ei >= 0; --ei) {
        const int l = ei / p.R, r = ei % p.R;
        const auto& bw = p.bw[l];
        const ExecSave& s = w.ex[ei];
        const BX bx = bexec(p, l, r);
        const float *wl = nullptr, *ww = nullptr;
        if (p.student && p.c.head_mix) { wl = PF(params, bx.cl); ww = PF(params, bx.cw); }
        // this execution's slots; a block's wgrads run once, after its first execution's gradients are there (r == 0)
        bf16_t* gb_f2 = w.gb_f2 + (int64_t)r * M * D;
        bf16_t* gb_pr = w.gb_pr + (int64_t)r * M * D;
        bf16_t* dbig = w.dbig + (int64_t)r * M * F;
        bf16_t* dqkv = w.dqkv + (int64_t)r * M * 3 * D;
        const int64_t MR = (int64_t)R * M;
        const ExecSave& s0 = w.ex[ei - r];                               // execution r = 0 of this block: base of the [R][M, .] operands
        // gradient arriving directly at this execution's output (feature-MSE terms): G += d_rep[ei], refresh the bf16 copy
        if (d_rep && d_rep[ei]) CK(dclip_axpy_f32(w.G, d_rep[ei], gb_f2, M * D, GR(bx.f2b), D, st));
        // MLP: x_out = x_mid + fc2(gelu(fc1(LN2(x_mid))))
        CK(dclip_gemm_nt(gb_f2, D, W + bw.fc2_t, D, dbig, F, M, F, D, 1.f, nullptr, DCLIP_ACT_MULAUX, s.z, nullptr, nullptr, 0, 0, 0, nullptr,
                         GR(bx.f1b), st));                                        // dz = (G W2) o gelu'(z) ; db1 += colsum(dz)
        if (r == 0 && GR(bx.f2w)) CK(dclip_gemm_tn_acc(w.gb_f2, D, s0.u, F, GR(bx.f2w), F, MR, D, F, wsplits(MR, D, F), w.tn_ws, w.tn_ws_bytes, st));
        if (r == 0 && GR(bx.f1w)) CK(dclip_gemm_tn_acc(w.dbig, F, s0.h2, D, GR(bx.f1w), D, MR, F, D, wsplits(MR, F, D), w.tn_ws, w.tn_ws_bytes, st));
        CK(gemm(dbig, F, W + bw.fc1_t, F, w.dh, D, M, D, F, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, st));
        CK(dclip_layernorm_bwd(w.dh, D, 0, (const float*)s.x_mid, D, nullptr, PF(params, bx.n2w), s.mean2, s.rstd2, w.G, D, gb_pr, D, GR(bx.n2w), GR(bx.n2b),
                               GR(bx.prb), M, D, st));
        // attention: x_mid = x_in + proj(attn(LN1(x_in)))
        if (r == 0 && GR(bx.prw)) CK(dclip_gemm_tn_acc(w.gb_pr, D, s0.ctx, D, GR(bx.prw), D, MR, D, D, wsplits(MR, D, D), w.tn_ws, w.tn_ws_bytes, st));
        bf16_t* dctx = w.dh;
        CK(gemm(gb_pr, D, W + bw.proj_t, D, dctx, D, M, D, D, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, st));
        const int blk = (wl && mix_attn(p, N)) ? 1 : 0;        // R and dS of the register-resident score stage are quad-blocked
        CK(dclip_attn_tn(s.Rm, dctx, D, dqkv + 2 * D, 3 * D, B, H, N, Np, hd, 1.f, blk, st));                    // dV = R^T dO
        if (wl && mix_attn(p, N)) {
            float* gl = GR(bx.cl) ? GR(bx.cl) : w.wg_dummy;
            float* gw = GR(bx.cw) ? GR(bx.cw) : w.wg_dummy + H * H;
            CK(dclip_attn_mix_bwd(s.qkv, 3 * D, dctx, D, wl, ww, s.stats, w.dS, gl, gw, w.mix_ws, w.mix_ws_bytes, B, H, N, Np, hd, scale, st));
        } else {
            CK(dclip_attn_nt(dctx, D, s.qkv + 2 * D, 3 * D, w.dR, 0, B, H, N, Np, hd, 1.f, st));               // dR = dO V^T
            CK(dclip_attn_softmax_bwd(w.dR, s.P, s.S, 0, wl, ww, w.dS, wl ? GR(bx.cl) : nullptr,
                                      wl ? GR(bx.cw) : nullptr, B, H, N, Np, st));
        }
        CK(dclip_attn_nn(w.dS, s.qkv + D, 3 * D, dqkv, 3 * D, B, H, N, Np, hd, scale, blk, st));                 // dQ = dS K
        CK(dclip_attn_tn(w.dS, s.qkv, 3 * D, dqkv + D, 3 * D, B, H, N, Np, hd, scale, blk, st));                 // dK = dS^T Q
        if (r == 0 && GR(bx.qkvw)) CK(dclip_gemm_tn_acc(w.dqkv, 3 * D, s0.h1, D, GR(bx.qkvw), D, MR, 3 * D, D, wsplits(MR, 3 * D, D), w.tn_ws, w.tn_ws_bytes, st));
        if (r == 0 && params[bx.qkvb] && GR(bx.qkvb)) CK(dclip_colsum_acc(w.dqkv, 3 * D, GR(bx.qkvb), MR, 3 * D, st));
        CK(gemm(dqkv, 3 * D, W + bw.qkv_t, 3 * D, w.dh, D, M, D, 3 * D, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, st));
        // the bf16 residual gradient leaving this execution is the fc2 operand of the previous one (slot of its repeat index)
        bf16_t* gb_next = ei > 0 ? w.gb_f2 + (int64_t)((ei - 1) % R) * M * D : w.Gb;
        CK(dclip_layernorm_bwd(w.dh, D, 0, (const float*)w.X[ei], D, nullptr, PF(params, bx.n1w), s.mean1, s.rstd1, w.G, D, gb_next, D, GR(bx.n1w), GR(bx.n1b),
                               ei > 0 ? GR(bexec(p, (ei - 1) / p.R, 0).f2b) : nullptr, M, D, st));
        // block l's gradients (shared weights: both repeats; its fc2 bias also collects from block l + 1's first LayerNorm
        // backward, which ran earlier) are complete after its first execution's backward: bucket 1 + (L - 1 - l)
        if (r == 0 && on_bucket) on_bucket(cb_user, 1 + (p.L - 1 - l));
    }

    // ---- embedding ---------------------------------------------------------------------------------------------
    const bool clip_image = !p.student && p.image;       // the exported embedding of a CLIP image tower is taken BEFORE ln_pre (_common.py:204-208)
    if (d_emb && !clip_image) CK(dclip_axpy_f32(w.G, d_emb, w.Gb, M * D, nullptr, D, st));
    if (hipMemsetAsync(w.tok_sum, 0, (size_t)N * D * 4, hs) != hipSuccess) { dclip_set_error("dclip_encoder_backward: memset failed"); return DCLIP_ELAUNCH; }
    if (clip_image) {        // grads: 0 conv1 w, 1 class_embedding, 2 positional_embedding, 3 ln_pre w, 4 ln_pre b
        // x = ln_pre(x0), no bypass: the gradient of x0 is LN'(G) alone, accumulated into a cleared buffer
        if (hipMemsetAsync(w.G0, 0, (size_t)M * D * 4, hs) != hipSuccess) { dclip_set_error("dclip_encoder_backward: memset failed"); return DCLIP_ELAUNCH; }
        CK(dclip_layernorm_bwd(w.G, D, 1, (const float*)w.x0, D, nullptr, PF(params, 3), w.mean0, w.rstd0, w.G0, D, w.Gb, D, GR(3), GR(4), nullptr, M, D, st));
        if (d_emb) CK(dclip_axpy_f32(w.G0, d_emb, w.Gb, M * D, nullptr, D, st));
        if (GR(0)) CK(dclip_gemm_tn_acc(w.Gb, D, ext_patches ? ext_patches : w.patches, p.K, GR(0), p.K, M, D, p.K, wsplits(M, D, p.K), w.tn_ws, w.tn_ws_bytes, st));
        if (GR(1) || GR(2)) {
            CK(dclip_batch_sum_acc(w.G0, w.tok_sum, B, N, D, st));
            CK(dclip_token_table_bwd(w.tok_sum, GR(2), GR(1), nullptr, N, D, 1, st));
        }
    } else if (p.image) {    // grads: 0 conv w, 1 conv b, 2 cls, 3 pos
        if (GR(0)) CK(dclip_gemm_tn_acc(w.Gb, D, ext_patches ? ext_patches : w.patches, p.K, GR(0), p.K, M, D, p.K, wsplits(M, D, p.K), w.tn_ws, w.tn_ws_bytes, st));
        if (GR(1) || GR(2) || GR(3)) {
            CK(dclip_batch_sum_acc(w.G, w.tok_sum, B, N, D, st));
            CK(dclip_token_table_bwd(w.tok_sum, GR(3), GR(2), GR(1), N, D, 1, st));
        }
    } else if (p.compressed) {   // grads: 0 table, 1 linear w, 2 linear b, 3 pos
        const int64_t rk = p.c.embed_rank;
        if (GR(1)) CK(dclip_gemm_tn_acc(w.Gb, D, w.patches, rk, GR(1), rk, M, D, rk, wsplits(M, D, rk), w.tn_ws, w.tn_ws_bytes, st));
        if (GR(2) || GR(3)) {
            CK(dclip_batch_sum_acc(w.G, w.tok_sum, B, N, D, st));
            CK(dclip_token_table_bwd(w.tok_sum, GR(3), nullptr, GR(2), N, D, 0, st));
        }
        if (GR(0)) {
            CK(gemm(w.Gb, D, W + p.w_embed_t, D, w.demb, rk, M, rk, D, nullptr, 0, nullptr, nullptr, nullptr, 0, 1, 0, nullptr, st));
            CK(dclip_embed_scatter_add((const int64_t*)input, w.demb, 1, GR(0), M, rk, p.c.vocab, st));
        }
    } else {                     // grads: 0 table, 1 pos
        if (GR(0)) CK(dclip_embed_scatter_add((const int64_t*)input, w.G, 1, GR(0), M, D, p.c.vocab, st));
        if (GR(1)) {
            CK(dclip_batch_sum_acc(w.G, w.tok_sum, B, N, D, st));
            CK(dclip_token_table_bwd(w.tok_sum, GR(1), nullptr, nullptr, N, D, 0, st));
        }
    }
    if (on_bucket) on_bucket(cb_user, p.L + 1);          // embedding parameters: the last bucket
    return DCLIP_OK;
}

extern "C" int dclip_encoder_backward(const dclip_encoder* e, const void* input, int64_t B, const void* const* params,
                                      void* const* grads, const void* wcache, void* workspace, size_t ws_bytes,
                                      const float* d_last_representation, const float* const* d_rep, const float* d_emb,
                                      dclip_bucket_cb on_bucket, void* cb_user, void* st) {
    DCLIP_REQUIRE(input, "dclip_encoder_backward: null argument");
    return encoder_backward_impl(e, input, nullptr, B, params, grads, wcache, workspace, ws_bytes, d_last_representation, d_rep, d_emb,
                                 on_bucket, cb_user, st);
}

extern "C" int dclip_encoder_backward_patches(const dclip_encoder* e, const void* patches, int64_t B, const void* const* params,
                                              void* const* grads, const void* wcache, void* workspace, size_t ws_bytes,
                                              const float* d_last_representation, const float* const* d_rep, const float* d_emb,
                                              dclip_bucket_cb on_bucket, void* cb_user, void* st) {
    DCLIP_REQUIRE(patches, "dclip_encoder_backward_patches: null argument");
    return encoder_backward_impl(e, nullptr, (const bf16_t*)patches, B, params, grads, wcache, workspace, ws_bytes, d_last_representation,
                                 d_rep, d_emb, on_bucket, cb_user, st);
}

// Gradient buckets in the order the backward completes them (data-parallel exchange, SURVEY.md section 8e Collective 1):
// bucket 0 = final norm + head, 1 .. L = blocks L-1 .. 0, L + 1 = embedding parameters.  Each is a contiguous range of the
// canonical parameter order, hence a contiguous range of a flat gradient buffer laid out in that order.
extern "C" int32_t dclip_encoder_num_grad_buckets(const dclip_encoder* e) { return e ? e->p.L + 2 : -1; }
extern "C" int dclip_encoder_grad_bucket(const dclip_encoder* e, int32_t bucket, int32_t* first_param, int32_t* end_param) {
    DCLIP_REQUIRE(e && first_param && end_param, "dclip_encoder_grad_bucket: null argument");
    const Plan& p = e->p;
    DCLIP_REQUIRE(bucket >= 0 && bucket <= p.L + 1, "dclip_encoder_grad_bucket: bucket %d out of range 0..%d", bucket, p.L + 1);
    const int per_block = p.student ? P_PER_SBLOCK + p.R * P_PER_SREPEAT : P_PER_TBLOCK;
    if (bucket == 0) { *first_param = p.p_final; *end_param = p.n_params; }
    else if (bucket == p.L + 1) { *first_param = 0; *end_param = p.p_blocks; }
    else { const int l = p.L - bucket; *first_param = p.p_blocks + l * per_block; *end_param = *first_param + per_block; }
    return DCLIP_OK;
}
