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
// LDS traffic of ONE wave (the weight-gradient tiles are wave-private): the LDS queue of a wave is in order, the fence keeps the
// compiler from moving accesses across it
DEVFN void lds_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); }
DEVFN void block_sync() { __syncthreads(); }
DEVFN unsigned long long clock() { return __builtin_readcyclecounter(); }
DEVFN void sched_fence() { __builtin_amdgcn_sched_barrier(0); }    // no instruction moves across this point
// LDS-DMA: 16 bytes per lane from the lane's own global address to (wave-uniform dst) + lane * 16; completion is tracked by vmcnt
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
DEVFN void dma16(const void* src, char* dst) { __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)dst, 16, 0, 0); }
// all but the N youngest vector-memory operations of this wave are complete (loads, stores and LDS-DMA retire in issue order)
template <int N> DEVFN void dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
}  // namespace hw

#include "attn_mix_wave.h"

namespace {

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

template <int H, int HD>
__global__ __launch_bounds__(256) void attn_mix_bwd_kernel(amix::BwdArgs p) {
    using C = amix::Cfg<H, HD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    amix::bwd_wave<C>(p, blockIdx.x, gridDim.x, wave, 4, lane, smem);
}

// dW[which][g, h] += mul * sum over the workgroups' partial tiles
__global__ void attn_mix_wgrad_reduce_kernel(const float* __restrict__ partial, int nwg, int H, int HP, float* dWl, float* dWw) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 2 * H * H) return;
    const int which = idx / (H * H), gh = idx - which * H * H, g = gh / H, h = gh - g * H;
    const float* src = partial + (size_t)which * HP * HP + g * HP + h;
    float s = 0.f;
    for (int w = 0; w < nwg; ++w) s += src[(size_t)w * 2 * HP * HP];
    float* dst = which == 0 ? dWl : dWw;
    dst[gh] += s;
}

constexpr int BWD_MAX_WG = 256;
unsigned long long* g_fwd_stamps = nullptr;      // diagnostics only (tools/diag/attn_mix_prof.py)

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
extern "C" void dclip_attn_mix_debug_stamps(void* buf) { g_fwd_stamps = (unsigned long long*)buf; }

extern "C" size_t dclip_attn_mix_bwd_workspace_bytes(int64_t H) {
    const size_t HP = (size_t)((H + 15) / 16) * 16;
    return (size_t)BWD_MAX_WG * 2 * HP * HP * sizeof(float);
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
    static const int occ = [] { const char* e = getenv("DCLIP_MIX_OCC"); return e ? atoi(e) : 2; }();
    if (occ == 1) {
        MIX_DISPATCH(H, hd, hipLaunchKernelGGL((attn_mix_fwd_kernel<HH, HD_, 1>), grid, dim3(256), (size_t)4 * amix::fwd_lds_per_wave<CC>(),
                                               (hipStream_t)stream, p));
    } else {
        MIX_DISPATCH(H, hd, hipLaunchKernelGGL((attn_mix_fwd_kernel<HH, HD_, 2>), grid, dim3(256), (size_t)4 * amix::fwd_lds_per_wave<CC>(),
                                               (hipStream_t)stream, p));
    }
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
    DCLIP_REQUIRE(ws_bytes >= dclip_attn_mix_bwd_workspace_bytes(H), "dclip_attn_mix_bwd: workspace too small (%zu < %zu)", ws_bytes,
                  dclip_attn_mix_bwd_workspace_bytes(H));
    const int QT = (int)((N + 15) / 16);
    const int HP = (int)((H + 15) / 16) * 16;
    int blocks = (int)((B * QT + 3) / 4);
    if (blocks > BWD_MAX_WG) blocks = BWD_MAX_WG;        // persistent: one weight-gradient partial per workgroup
    amix::BwdArgs p{(const bf16_t*)qkv, (long)ld, (const bf16_t*)dO, (long)ldo, Wl, Ww, stats, (bf16_t*)dS, (float*)workspace, (int)B, (int)N, (int)Np, QT, scale};
    const double el = (double)B * H * N * Np;
    TraceScope tr(DCLIP_TRACE_ATTN, 8.0 * B * H * N * N * hd + 20.0 * el * H, 2.0 * el + 8.0 * B * N * H * hd, stream, (int)(B * H), (int)N, (int)hd, 8);
    MIX_DISPATCH(H, hd, hipLaunchKernelGGL((attn_mix_bwd_kernel<HH, HD_>), dim3(blocks), dim3(256), (size_t)4 * amix::bwd_lds_per_wave<CC>(),
                                           (hipStream_t)stream, p));
    hipLaunchKernelGGL(attn_mix_wgrad_reduce_kernel, dim3((unsigned)((2 * H * H + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, blocks, (int)H, HP, dWl, dWw);
    return dclip_check_launch("dclip_attn_mix_bwd");
}
