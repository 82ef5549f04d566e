// Head-mixed student attention, score stage, with the score tensors in registers and both head mixes on the matrix pipe.
//
//   reference: model/component/weight_share_model.py:101-125 (MiniAttention.forward)
//       S_h = scale * Q_h K_h^T ; A = conv_l(S) (1x1 conv over the head channel) ; P = softmax_j(A) ; R = conv_w(P) ; ctx_h = R_h V_h
//
// The algorithm (block-diagonal score MFMAs that leave lane group g4 with head 4s + g4, mixes as MFMAs whose B operand is packed
// straight from accumulator registers, softmax statistics as per-register running sums) lives in attn_mix_wave.h, which is also
// compiled for the host against a 64-lane emulation (tools/emu/) to check the index maps on the CPU.  This file binds it to gfx950:
//
//   dclip_attn_mix_fwd : one wave per (sample, 16 queries); pass 1 statistics, pass 2 P and R.  S, A, P never exist in memory; R
//                        (bf16) and the log-sum-exp rows are stored.
//   dclip_attn_mix_bwd : persistent waves; pass A delta = sum_j P dP (and dW_w), pass B dA, dS (and dW_l); dR = dO v^T formed on
//                        the fly by the same block-diagonal product.  dS (bf16) is stored; the weight gradients leave as ONE
//                        partial tile per workgroup and a second launch adds them up (no same-line atomics, run-to-run identical).
#include <math.h>
#include <stdlib.h>
#include "common.h"

#define DEVFN __device__ __forceinline__
#define DEVMEM __device__ __forceinline__
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

namespace hw {
DEVFN f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
DEVFN f32x4 mfma_f16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
DEVFN float exp2(float x) { return __builtin_amdgcn_exp2f(x); }
DEVFN float log2(float x) { return __builtin_amdgcn_logf(x); }
DEVFN bool any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }
// LDS traffic of ONE wave (the weight-gradient tiles are wave-private).  The DS queue of a wave is in order, so a ds_read issued
// after a ds_write sees the data of every lane; all that is needed is that the compiler keeps the program order.  (A workgroup-
// scope release fence would also emit s_waitcnt vmcnt(0): the full latency of the LDS-DMA request in flight for the next quad.)
DEVFN void lds_fence() { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
DEVFN void block_sync() { __syncthreads(); }
// 16-byte LDS fragment read the compiler does not see as an LDS access (no s_waitcnt of its own: pair it with lds_wait_frags)
template <int OFF>
DEVFN bf16x8 lds_read_frag(const char* p) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(uintptr_t)p), "n"(OFF));
    return v;
}
// at most N LDS operations issued after the reads of `f` are still outstanding -> `f` has landed (LDS returns in order); naming
// the registers read-write keeps every consumer behind this statement
template <int N, int GF>
DEVFN void lds_wait_frags(bf16x8 (&f)[GF]) {
    static_assert(GF == 4 || GF == 8, "fragment group of 4 or 8");
    if constexpr (GF == 8)
        asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "n"(N));
    else
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(N));
}
// the value lives in an AGPR from here on (MFMA B operands may be AGPRs; VGPRs stay free for what the VALU touches)
DEVFN void pin_acc(bf16x8& v) { asm volatile("" : "+a"(v)); }
DEVFN unsigned long long clock() { return __builtin_readcyclecounter(); }
// MFMA source registers must outlive the MFMA's issue by ~10 issue slots.  The matrix pipe accepts MFMAs faster than it starts
// them: an MFMA that queues behind two or three others reads its A / B registers only when its turn comes, and neither the hardware
// nor hipcc's hazard recogniser keeps a VALU instruction from overwriting them meanwhile (asynchronous LDS / memory returns arrive
// late enough).  Found with H = 8, hd = 32: the compiler reused the packed dA operand of the LAST of four back-to-back mix MFMAs
// three instructions after it — wrong dS for the fourth key of every quad but the (peeled) first, deterministically, with no
// tool complaining (tools/diag/mix_fuzz.py).  keep_alive() pins operands built by the VALU until a point well past the group,
// at no cost in instructions; mfma_src_guard() is the blunt form (12 idle slots) where the operands cannot be named.
template <class T8> DEVFN void keep_alive(const T8 (&f)[4]) { asm volatile("" :: "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3])); }
template <class T8> DEVFN void keep_alive(const T8& f) { asm volatile("" :: "v"(f)); }
DEVFN void mfma_src_guard() { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 7\n\ts_nop 3" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
// no instruction moves across this point.  Required, not a tuning knob: the hand-placed ds_read / s_waitcnt pairs around the ring
// are only ordered against the MFMAs that consume them by these fences (a build without them fails test_kernels_gpu).
DEVFN void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// LDS-DMA: 16 bytes per lane from the lane's own global address to (wave-uniform dst) + lane * 16; completion is tracked by vmcnt
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
DEVFN void dma16(const void* src, char* dst) { __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)dst, 16, 0, 0); }
// all but the N youngest vector-memory operations of this wave are complete (loads, stores and LDS-DMA retire in issue order)
template <int N> DEVFN void dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
}  // namespace hw

#include "attn_mix_wave.h"

namespace {

constexpr int BWD_MAX_WG = 256;

// OCC = waves per SIMD the register allocation is held to (2: two workgroups per CU hide each other's latencies, at 256 registers)
template <int H, int HD, int OCC>
__global__ __launch_bounds__(256, OCC) void attn_mix_fwd_kernel(amix::FwdArgs p) {
    using C = amix::Cfg<H, HD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * 4 + wave;
    if (item >= p.B * p.QT) return;
    const int b = item / p.QT, it = item - b * p.QT;
    const unsigned long long te = p.stamps ? hw::clock() : 0;
    char* lds = smem + wave * amix::fwd_lds_per_wave<C>();
    amix::zero_block_init<C>(lds, lane);
    amix::FwdWeights<C> w;
    amix::fwd_load_weights<C>(p, lane, w);
    if (p.stamps && lane == 0) {
        p.stamps[12 * (long)item + 8] = hw::clock() - te;                          // zero block + weight fragments
        p.stamps[12 * (long)item + 10] = __builtin_amdgcn_s_memrealtime();        // 100 MHz: wave start
    }
    amix::fwd_item<C>(p, b, it, lane, w, lds);
    if (p.stamps && lane == 0) p.stamps[12 * (long)item + 11] = __builtin_amdgcn_s_memrealtime();
}

template <int H, int HD, bool PASS_B>
__global__ __launch_bounds__(256) void attn_mix_bwd_kernel(amix::BwdArgs p) {
    using C = amix::Cfg<H, HD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    amix::bwd_wave<C, PASS_B>(p, blockIdx.x, gridDim.x, wave, 4, lane, smem);
}

// dW[which][g, h] += sum over the workgroups' partial tiles: one 256-thread block per output element, one partial per thread
// (fixed summation order: run-to-run identical)
__global__ __launch_bounds__(256) void attn_mix_wgrad_reduce_kernel(const float* __restrict__ partial, int nwg, int H, int HP, float* dWl, float* dWw) {
    __shared__ float part[4];
    const int idx = blockIdx.x;                       // (which, g, h)
    const int which = idx / (H * H), gh = idx - which * H * H, g = gh / H, h = gh - g * H;
    const int w = threadIdx.x;
    float v = w < nwg ? partial[((size_t)w * 2 + which) * HP * HP + g * HP + h] : 0.f;
    v = wave_sum(v);
    if ((w & 63) == 0) part[w >> 6] = v;
    __syncthreads();
    if (w == 0) {
        float* dst = which == 0 ? dWl : dWw;
        dst[gh] += (part[0] + part[1]) + (part[2] + part[3]);
    }
}

unsigned long long* g_fwd_stamps = nullptr;      // diagnostics only (tools/diag/attn_mix_prof.py)
unsigned long long* g_bwd_stamps[2] = {nullptr, nullptr};

}  // namespace

#define MIX_DISPATCH(Hv, HDv, ...)                                                     \
    switch ((Hv) * 100 + (HDv)) {                                                      \
        case 232: { constexpr int HH = 2, HD_ = 32; typedef amix::Cfg<HH, HD_> CC; __VA_ARGS__; break; }              \
        case 264: { constexpr int HH = 2, HD_ = 64; typedef amix::Cfg<HH, HD_> CC; __VA_ARGS__; break; }              \
        case 432: { constexpr int HH = 4, HD_ = 32; typedef amix::Cfg<HH, HD_> CC; __VA_ARGS__; break; }              \
        case 464: { constexpr int HH = 4, HD_ = 64; typedef amix::Cfg<HH, HD_> CC; __VA_ARGS__; break; }              \
        case 832: { constexpr int HH = 8, HD_ = 32; typedef amix::Cfg<HH, HD_> CC; __VA_ARGS__; break; }              \
        case 864: { constexpr int HH = 8, HD_ = 64; typedef amix::Cfg<HH, HD_> CC; __VA_ARGS__; break; }              \
        case 1232: { constexpr int HH = 12, HD_ = 32; typedef amix::Cfg<HH, HD_> CC; __VA_ARGS__; break; }            \
        case 1264: { constexpr int HH = 12, HD_ = 64; typedef amix::Cfg<HH, HD_> CC; __VA_ARGS__; break; }            \
        case 2432: { constexpr int HH = 24, HD_ = 32; typedef amix::Cfg<HH, HD_> CC; __VA_ARGS__; break; }            \
        default: dclip_set_error("attn_mix: no instantiation for H=%d hd=%d", (int)(Hv), (int)(HDv)); return DCLIP_EINVAL; \
    }

extern "C" int dclip_attn_mix_supported(int64_t H, int64_t N, int64_t hd) {
    const bool shape = ((H == 2 || H == 4 || H == 8 || H == 12) && (hd == 32 || hd == 64)) || (H == 24 && hd == 32);
    // width % 128: a quad of token rows is a whole number of 1-KiB LDS-DMA pieces
    return shape && H * hd <= 1024 && (H * hd) % 128 == 0 && N >= 1 && N <= 128;
}

// diagnostics: subsequent dclip_attn_mix_fwd launches of this process write 12 counts per (sample, 16-query tile) to `buf`
// (u64 [B * ceil(N / 16)][12]: pass 1 total / ring wait / score MFMAs / per-key stage, then the same for pass 2); nullptr = off
// bwd_a / bwd_b: u64 [B * ceil(N / 16)][8] for the two launches of dclip_attn_mix_bwd (total, ring wait, S + dR, per-key stage, dW product)
extern "C" void dclip_attn_mix_debug_stamps(void* fwd, void* bwd_a, void* bwd_b) {
    g_fwd_stamps = (unsigned long long*)fwd;
    g_bwd_stamps[0] = (unsigned long long*)bwd_a;
    g_bwd_stamps[1] = (unsigned long long*)bwd_b;
}

static size_t partial_bytes(int64_t H) {
    const size_t HP = (size_t)((H + 15) / 16) * 16;
    return (size_t)BWD_MAX_WG * 2 * HP * HP * sizeof(float);
}
// per-workgroup weight-gradient partials [256][2][HP][HP] f32, then delta [B, H, N] f32
extern "C" size_t dclip_attn_mix_bwd_workspace_bytes(int64_t B, int64_t H, int64_t N) {
    return partial_bytes(H) + (((size_t)B * H * N * sizeof(float) + 255) & ~(size_t)255);
}

extern "C" int dclip_attn_mix_fwd(const void* qkv, int64_t ld, const float* Wl, const float* Ww, void* R, float* stats, int64_t B,
                                  int64_t H, int64_t N, int64_t Np, int64_t hd, float scale, void* stream) {
    DCLIP_REQUIRE(qkv && Wl && Ww && R && stats && B > 0, "dclip_attn_mix_fwd: null / empty argument");
    DCLIP_REQUIRE(dclip_attn_mix_supported(H, N, hd), "dclip_attn_mix_fwd: unsupported shape H=%ld N=%ld hd=%ld", (long)H, (long)N, (long)hd);
    DCLIP_REQUIRE(Np == ((N + 7) & ~(int64_t)7) && ld % 8 == 0 && ld >= 3 * H * hd && ((uintptr_t)qkv % 16) == 0 && ((uintptr_t)R % 16) == 0,
                  "dclip_attn_mix_fwd: misaligned buffers (Np = round_up(N, 8), 16-byte aligned qkv rows and R)");
    const int QT = (int)((N + 15) / 16);
    amix::FwdArgs p{(const bf16_t*)qkv, (long)ld, Wl, Ww, (bf16_t*)R, stats, (int)B, (int)N, (int)Np, QT, scale, g_fwd_stamps};
    const dim3 grid((unsigned)((B * QT + 3) / 4));
    const double el = (double)B * H * N * Np;
    TraceScope tr(DCLIP_TRACE_ATTN, 4.0 * B * H * N * N * hd + 8.0 * el * H, 2.0 * el + 4.0 * B * N * H * hd, stream, (int)(B * H), (int)N, (int)hd, 7);
    MIX_DISPATCH(H, hd, hipLaunchKernelGGL((attn_mix_fwd_kernel<HH, HD_, 2>), grid, dim3(256), (size_t)4 * amix::fwd_lds_per_wave<CC>(),
                                           (hipStream_t)stream, p));
    return dclip_check_launch("dclip_attn_mix_fwd");
}

extern "C" int dclip_attn_mix_bwd(const void* qkv, int64_t ld, const void* dO, int64_t ldo, const float* Wl, const float* Ww,
                                  const float* stats, void* dS, float* dWl, float* dWw, void* workspace, size_t ws_bytes, int64_t B,
                                  int64_t H, int64_t N, int64_t Np, int64_t hd, float scale, void* stream) {
    DCLIP_REQUIRE(qkv && dO && Wl && Ww && stats && dS && dWl && dWw && workspace && B > 0, "dclip_attn_mix_bwd: null / empty argument");
    DCLIP_REQUIRE(dclip_attn_mix_supported(H, N, hd), "dclip_attn_mix_bwd: unsupported shape H=%ld N=%ld hd=%ld", (long)H, (long)N, (long)hd);
    DCLIP_REQUIRE(Np == ((N + 7) & ~(int64_t)7) && ld % 8 == 0 && ldo % 8 == 0 && ld >= 3 * H * hd && ldo >= H * hd && ((uintptr_t)qkv % 16) == 0 &&
                      ((uintptr_t)dO % 16) == 0 && ((uintptr_t)dS % 16) == 0 && ((uintptr_t)workspace % 16) == 0,
                  "dclip_attn_mix_bwd: misaligned buffers");
    DCLIP_REQUIRE(ws_bytes >= dclip_attn_mix_bwd_workspace_bytes(B, H, N), "dclip_attn_mix_bwd: workspace too small (%zu < %zu)", ws_bytes,
                  dclip_attn_mix_bwd_workspace_bytes(B, H, N));
    const int QT = (int)((N + 15) / 16);
    const int HP = (int)((H + 15) / 16) * 16;
    int blocks = (int)((B * QT + 3) / 4);
    if (blocks > BWD_MAX_WG) blocks = BWD_MAX_WG;        // persistent: one weight-gradient partial per workgroup
    amix::BwdArgs p{(const bf16_t*)qkv, (long)ld, (const bf16_t*)dO, (long)ldo, Wl, Ww, stats, (bf16_t*)dS, (float*)workspace,
                    (float*)((char*)workspace + partial_bytes(H)), (int)B, (int)N, (int)Np, QT, scale, g_bwd_stamps[0]};
    const double el = (double)B * H * N * Np;
    TraceScope tr(DCLIP_TRACE_ATTN, 8.0 * B * H * N * N * hd + 20.0 * el * H, 2.0 * el + 8.0 * B * N * H * hd, stream, (int)(B * H), (int)N, (int)hd, 8);
    // pass A: delta = sum_j P dP, dW_w partials ; pass B: dS, dW_l partials
    MIX_DISPATCH(H, hd, hipLaunchKernelGGL((attn_mix_bwd_kernel<HH, HD_, false>), dim3(blocks), dim3(256), (size_t)4 * amix::bwd_lds_per_wave<CC>(),
                                           (hipStream_t)stream, p));
    p.stamps = g_bwd_stamps[1];
    MIX_DISPATCH(H, hd, hipLaunchKernelGGL((attn_mix_bwd_kernel<HH, HD_, true>), dim3(blocks), dim3(256), (size_t)4 * amix::bwd_lds_per_wave<CC>(),
                                           (hipStream_t)stream, p));
    static_assert(BWD_MAX_WG <= 256, "one partial per thread of the reduction block");
    hipLaunchKernelGGL(attn_mix_wgrad_reduce_kernel, dim3((unsigned)(2 * H * H)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, blocks, (int)H, HP, dWl, dWw);
    return dclip_check_launch("dclip_attn_mix_bwd");
}
