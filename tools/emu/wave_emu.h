// 64-lane wave emulation for the host: runs the per-wave device code of distillclip_amd/csrc/*_wave.h on the CPU so that the lane /
// register index maps of the MFMA-based kernels can be checked without a GPU (tools/emu/README.md).  TEST INFRASTRUCTURE ONLY:
// nothing under distillclip_amd/ includes this file, and the product path has no CPU fallback.
//
// Every lane of a workgroup is a ucontext coroutine; a scheduler resumes them round-robin.  A lane runs until it reaches a wave
// collective (MFMA, vote, LDS fence) or a workgroup barrier, deposits its operands and yields; the last lane of the wave to arrive
// evaluates the collective for all 64 lanes.  Plain loads / stores go to host memory, LDS is a per-workgroup byte array.
//
// MFMA semantics emulated (v_mfma_f32_16x16x32_{bf16,f16}; cdna_hip_programming.md section 3):
//   A fragment of lane l = A[row = l & 15][k = 8 (l >> 4) + 0..7], B fragment = B[k = 8 (l >> 4) + 0..7][col = l & 15],
//   accumulator register r of lane l = D[row = 4 (l >> 4) + r][col = l & 15], f32 accumulation.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <deque>
#include <functional>
#include <vector>

#define DCLIP_EMU 1
#define DEVFN static inline
#define DEVMEM inline

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

namespace emu {

struct WaveSlots {
    alignas(16) char a[64][16];
    alignas(16) char b[64][16];
    f32x4 c[64], res[64];
    bool vote_in[64];
    bool vote_out = false;
    int count = 0, gen = 0, kind = -1;
};

struct PendingDma { char* dst; char data[16]; };

struct Group {
    std::vector<std::deque<PendingDma>> dmaq;       // per lane, in issue order: LDS-DMA that has not "landed" yet
    int nthreads = 0;
    std::vector<ucontext_t> ctx;
    std::vector<std::vector<char>> stacks;
    std::vector<char> done;
    std::vector<WaveSlots> waves;
    ucontext_t sched;
    int cur = 0;
    int bar_count = 0, bar_gen = 0;
    std::function<void(int)> body;
    long collectives = 0;
};

static thread_local Group* G = nullptr;

static inline int tid() { return G->cur; }
static inline void yield() { swapcontext(&G->ctx[G->cur], &G->sched); }

static void trampoline() {
    Group* g = G;
    const int t = g->cur;
    g->body(t);
    g->done[t] = 1;
    swapcontext(&g->ctx[t], &g->sched);
}

// run `body(thread id)` for one workgroup of nthreads (a multiple of 64); lds is shared by the workgroup
static inline void run_group(int nthreads, const std::function<void(int)>& body, size_t stack_bytes = 512 * 1024) {
    Group g;
    g.nthreads = nthreads;
    g.ctx.resize(nthreads);
    g.stacks.resize(nthreads);
    g.done.assign(nthreads, 0);
    g.waves.resize(nthreads / 64);
    g.dmaq.resize(nthreads);
    g.body = body;
    G = &g;
    for (int t = 0; t < nthreads; ++t) {
        g.stacks[t].resize(stack_bytes);
        getcontext(&g.ctx[t]);
        g.ctx[t].uc_stack.ss_sp = g.stacks[t].data();
        g.ctx[t].uc_stack.ss_size = stack_bytes;
        g.ctx[t].uc_link = nullptr;
        makecontext(&g.ctx[t], trampoline, 0);
    }
    for (;;) {
        bool any = false;
        for (int t = 0; t < nthreads; ++t) {
            if (g.done[t]) continue;
            any = true;
            g.cur = t;
            swapcontext(&g.sched, &g.ctx[t]);
        }
        if (!any) break;
    }
    G = nullptr;
}

enum { K_MFMA_BF16 = 1, K_MFMA_F16 = 2, K_VOTE = 3, K_FENCE = 4 };

static inline void check_kind(WaveSlots& w, int kind) {
    if (w.count == 0) w.kind = kind;
    else if (w.kind != kind) { fprintf(stderr, "emu: lanes of one wave diverged at a collective (%d vs %d)\n", w.kind, kind); abort(); }
}

template <class T>
static inline void mfma_all(WaveSlots& w) {
    // D[row][col] = C[row][col] + sum_k A[row][k] B[k][col]
    float A[16][32], B[32][16];
    for (int l = 0; l < 64; ++l) {
        const T* a = (const T*)w.a[l];
        const T* b = (const T*)w.b[l];
        for (int j = 0; j < 8; ++j) {
            A[l & 15][8 * (l >> 4) + j] = (float)a[j];
            B[8 * (l >> 4) + j][l & 15] = (float)b[j];
        }
    }
    for (int l = 0; l < 64; ++l) {
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * (l >> 4) + r, col = l & 15;
            float acc = w.c[l][r];
            for (int k = 0; k < 32; ++k) acc = fmaf(A[row][k], B[k][col], acc);
            w.res[l][r] = acc;
        }
    }
}

static inline f32x4 mfma(int kind, const void* a, const void* b, f32x4 c) {
    Group* g = G;
    const int t = g->cur, l = t & 63;
    WaveSlots& w = g->waves[t >> 6];
    check_kind(w, kind);
    memcpy(w.a[l], a, 16);
    memcpy(w.b[l], b, 16);
    w.c[l] = c;
    const int gen = w.gen;
    if (++w.count == 64) {
        if (kind == K_MFMA_BF16) mfma_all<__bf16>(w); else mfma_all<_Float16>(w);
        w.count = 0;
        ++w.gen;
        ++g->collectives;
    } else {
        while (w.gen == gen) yield();
    }
    return w.res[l];
}

static inline bool vote_any(bool p) {
    Group* g = G;
    const int t = g->cur, l = t & 63;
    WaveSlots& w = g->waves[t >> 6];
    check_kind(w, K_VOTE);
    w.vote_in[l] = p;
    const int gen = w.gen;
    if (++w.count == 64) {
        bool r = false;
        for (int i = 0; i < 64; ++i) r = r || w.vote_in[i];
        w.vote_out = r;
        w.count = 0;
        ++w.gen;
    } else {
        while (w.gen == gen) yield();
    }
    return w.vote_out;
}

static inline void wave_sync() {
    Group* g = G;
    const int t = g->cur;
    WaveSlots& w = g->waves[t >> 6];
    check_kind(w, K_FENCE);
    const int gen = w.gen;
    if (++w.count == 64) { w.count = 0; ++w.gen; }
    else while (w.gen == gen) yield();
}

// LDS-DMA (global_load_lds, 16 bytes per lane): the destination is wave-uniform base + lane * 16.  The bytes land only at a
// dma_wait that leaves fewer operations outstanding -- reading the stage earlier returns the OLD LDS content, as on the hardware.
static inline void dma16(const void* src, char* dst_base) {
    Group* g = G;
    const int t = g->cur;
    PendingDma d;
    d.dst = dst_base + (t & 63) * 16;
    memcpy(d.data, src, 16);
    g->dmaq[t].push_back(d);
}
static inline void dma_wait(int leave) {
    Group* g = G;
    auto& q = g->dmaq[g->cur];
    while ((int)q.size() > leave) { memcpy(q.front().dst, q.front().data, 16); q.pop_front(); }
    wave_sync();
}

static inline void block_sync() {
    Group* g = G;
    const int gen = g->bar_gen;
    if (++g->bar_count == g->nthreads) { g->bar_count = 0; ++g->bar_gen; }
    else while (g->bar_gen == gen) yield();
}

}  // namespace emu

namespace hw {
static inline f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) { return emu::mfma(emu::K_MFMA_BF16, &a, &b, c); }
static inline f32x4 mfma_f16(f16x8 a, f16x8 b, f32x4 c) { return emu::mfma(emu::K_MFMA_F16, &a, &b, c); }
static inline float exp2(float x) { return exp2f(x); }
static inline float log2(float x) { return log2f(x); }
static inline bool any(bool p) { return emu::vote_any(p); }
static inline void lds_fence() { emu::wave_sync(); }
static inline void block_sync() { emu::block_sync(); }
static inline void sched_fence() {}
static inline void pin_acc(bf16x8&) {}
template <int OFF> static inline bf16x8 lds_read_frag(const char* p) { return *(const bf16x8*)(p + OFF); }
template <int N, int GF> static inline void lds_wait_frags(bf16x8 (&)[GF]) {}
static inline unsigned long long clock() { return 0; }
template <class T8> static inline void keep_alive(const T8 (&)[4]) {}
template <class T8> static inline void keep_alive(const T8&) {}
static inline void mfma_src_guard() {}
static inline void dma16(const void* src, char* dst) { emu::dma16(src, dst); }
template <int N> static inline void dma_wait() { emu::dma_wait(N); }
}  // namespace hw
