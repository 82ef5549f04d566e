// LayerNorm forward / backward (fp32 statistics, eps inside the sqrt), HBM-bound.
//   reference: model/component/_common.py:14-20 (fp32 LayerNorm), nn.LayerNorm in weight_share_model.py:239,
//   call sites _common.py:123,125,208,210 ; text_encoder.py:69 ; weight_share_model.py:181,183,363,503.
// One wave per row, the row lives in registers (float4 chunks, D <= 1024), two-pass mean / variance.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int LN_MAXV = 4;   // float4 chunks per lane: D <= 64 * 4 * 4 = 1024

// X_F16: the rows are fp16 — the frozen teacher's residual stream, stored in the 16-bit type the reference's `precision: 16` autocast
// keeps it in (_common.py:14-20: the custom LayerNorm computes in fp32 and returns the input's type); statistics stay f32
template <int NV, int OUT, bool X_F16>      // OUT: 0 bf16, 1 f32, 2 f16
__global__ __launch_bounds__(256) void ln_fwd_kernel(const void* __restrict__ x, int64_t ldx, const int* __restrict__ ridx,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     void* __restrict__ y, int64_t ldy, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int M, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int64_t src = ridx ? ridx[row] : row;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = float4{0.f, 0.f, 0.f, 0.f};
        if (c < D) {
            if (X_F16) {
                const f16x4 h = *(const f16x4*)((const _Float16*)x + src * ldx + c);
                v[i] = float4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
            } else {
                v[i] = *(const float4*)((const float*)x + src * ldx + c);
            }
        }
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mu = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rs = rsqrtf(wave_sum(q) / D + eps);
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
            const float o0 = (v[i].x - mu) * rs * g.x + b.x, o1 = (v[i].y - mu) * rs * g.y + b.y;
            const float o2 = (v[i].z - mu) * rs * g.z + b.z, o3 = (v[i].w - mu) * rs * g.w + b.w;
            if (OUT == 1) {
                *(float4*)((float*)y + (int64_t)row * ldy + c) = float4{o0, o1, o2, o3};
            } else if (OUT == 2) {
                *(f16x4*)((_Float16*)y + (int64_t)row * ldy + c) = f16x4{(_Float16)o0, (_Float16)o1, (_Float16)o2, (_Float16)o3};
            } else {
                bf16x4 o = {f2bf(o0), f2bf(o1), f2bf(o2), f2bf(o3)};
                *(bf16x4*)((bf16_t*)y + (int64_t)row * ldy + c) = o;
            }
        }
    }
}

// backward: dx_acc[row] += rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
//           dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy       (per-wave register partials -> LDS -> atomics)
template <int NV, bool DY_F32>
__global__ __launch_bounds__(512) void ln_bwd_kernel(const void* __restrict__ dy, int64_t lddy, const float* __restrict__ x,
                                                     int64_t ldx, const int* __restrict__ ridx,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, float* __restrict__ dx_acc,
                                                     int64_t lddx, bf16_t* __restrict__ dx_bf16, int64_t lddb,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ colsum, int M, int D) {
    // 8 waves per block (the register budget allows 2 waves per SIMD = one such block per CU): the block's parameter-gradient
    // partials are summed through LDS and leave as ONE set of 3 D atomics — all blocks add to the same 3 D / 32 cache lines, and
    // same-line atomics serialise (4-wave blocks spent 10-17 % of the kernel there: tools/diag/ln_bench.py with the atomics off)
    constexpr int NW = 8, DP = NV * 256;
    extern __shared__ float red_[];                      // [3][NW][DP]
    float (*red)[NW][DP] = (float (*)[NW][DP])red_;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 dg[NV], db[NV], gm[NV], cs[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        dg[i] = db[i] = cs[i] = float4{0.f, 0.f, 0.f, 0.f};
        const int c = (i * 64 + lane) * 4;
        gm[i] = c < D ? *(const float4*)(gamma + c) : float4{0.f, 0.f, 0.f, 0.f};
    }
    // Software pipeline over rows: the loads of the NEXT row are issued before the stores of the current one.  vmcnt retires
    // loads and stores in one in-order queue, so a row whose loads follow the previous row's stores (the round-1 loop: load two
    // rows, reduce, store two rows, repeat) cannot start before those stores are acknowledged.
    struct Raw {
        float4 x[NV], a[NV], d[NV];
        float mu, rs;
        int64_t src;
        int row;
    };
    auto issue = [&](int row, Raw& r) {
        r.row = row;
        r.src = ridx ? ridx[row] : row;
        r.mu = mean[row];
        r.rs = rstd[row];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            r.x[i] = r.a[i] = r.d[i] = float4{0.f, 0.f, 0.f, 0.f};
            if (c < D) {
                r.x[i] = *(const float4*)(x + r.src * ldx + c);
                r.a[i] = *(const float4*)(dx_acc + r.src * lddx + c);
                if (DY_F32) r.d[i] = *(const float4*)((const float*)dy + (int64_t)row * lddy + c);
                else {
                    const bf16x4 t = *(const bf16x4*)((const bf16_t*)dy + (int64_t)row * lddy + c);
                    union { bf16x4 h; float2 f; } u; u.h = t;
                    r.d[i].x = u.f.x; r.d[i].y = u.f.y;            // raw bits, expanded in process()
                }
            }
        }
    };
    auto process = [&](const Raw& r) {
        float4 xh[NV], g[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            xh[i] = g[i] = float4{0.f, 0.f, 0.f, 0.f};
            if (c < D) {
                float4 d;
                if (DY_F32) d = r.d[i];
                else {
                    union { bf16x4 h; float2 f; } u; u.f = float2{r.d[i].x, r.d[i].y};
                    d = float4{bf2f(u.h[0]), bf2f(u.h[1]), bf2f(u.h[2]), bf2f(u.h[3])};
                }
                const float4 xv = r.x[i];
                xh[i] = float4{(xv.x - r.mu) * r.rs, (xv.y - r.mu) * r.rs, (xv.z - r.mu) * r.rs, (xv.w - r.mu) * r.rs};
                g[i] = float4{d.x * gm[i].x, d.y * gm[i].y, d.z * gm[i].z, d.w * gm[i].w};
                s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
                s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
                dg[i].x += d.x * xh[i].x; dg[i].y += d.y * xh[i].y; dg[i].z += d.z * xh[i].z; dg[i].w += d.w * xh[i].w;
                db[i].x += d.x; db[i].y += d.y; db[i].z += d.z; db[i].w += d.w;
            }
        }
        const float c1 = wave_sum(s1) / D, c2 = wave_sum(s2) / D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < D) {
                float* o = dx_acc + r.src * lddx + c;
                float4 a = r.a[i];
                a.x += r.rs * (g[i].x - c1 - xh[i].x * c2);
                a.y += r.rs * (g[i].y - c1 - xh[i].y * c2);
                a.z += r.rs * (g[i].z - c1 - xh[i].z * c2);
                a.w += r.rs * (g[i].w - c1 - xh[i].w * c2);
                *(float4*)o = a;
                cs[i].x += a.x; cs[i].y += a.y; cs[i].z += a.z; cs[i].w += a.w;
                if (dx_bf16) {
                    bf16x4 bq = {f2bf(a.x), f2bf(a.y), f2bf(a.z), f2bf(a.w)};
                    *(bf16x4*)(dx_bf16 + r.src * lddb + c) = bq;
                }
            }
        }
    };
    const int stride = gridDim.x * NW;
    Raw ra, rb;
    int row = blockIdx.x * NW + wave;
    if (row < M) issue(row, ra);
    while (row < M) {
        int nrow = row + stride;
        if (nrow < M) issue(nrow, rb);
        process(ra);
        row = nrow;
        if (row >= M) break;
        nrow = row + stride;
        if (nrow < M) issue(nrow, ra);
        process(rb);
        row = nrow;
    }
    // block reduction of the parameter gradients, then one atomic per column per block
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        *(float4*)&red[0][wave][c] = dg[i];
        *(float4*)&red[1][wave][c] = db[i];
        *(float4*)&red[2][wave][c] = cs[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 64 * NW) {
        float a = 0.f, b = 0.f, k = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { a += red[0][w][c]; b += red[1][w][c]; k += red[2][w][c]; }
        if (dgamma) unsafeAtomicAdd(dgamma + c, a);
        if (dbeta) unsafeAtomicAdd(dbeta + c, b);
        if (colsum) unsafeAtomicAdd(colsum + c, k);
    }
}

}  // namespace

#define LN_DISPATCH(NVV, ...)                     \
    switch (NVV) {                                \
        case 1: { constexpr int NV = 1; __VA_ARGS__; break; } \
        case 2: { constexpr int NV = 2; __VA_ARGS__; break; } \
        case 3: { constexpr int NV = 3; __VA_ARGS__; break; } \
        default: { constexpr int NV = 4; __VA_ARGS__; break; } \
    }

static int ln_fwd_launch(const void* x, int x_f16, int64_t ldx, const int32_t* row_index, const float* gamma, const float* beta, void* y,
                         int64_t ldy, int out, float* mean, float* rstd, int64_t M, int64_t D, float eps, void* stream, const char* who) {
    DCLIP_REQUIRE(x && gamma && beta && y, "%s: null operand", who);
    DCLIP_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 1024, "%s: need 0 < D <= 1024, D %% 4 == 0 (D=%ld)", who, (long)D);
    DCLIP_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ((uintptr_t)x % (x_f16 ? 8 : 16)) == 0, "%s: row strides must be multiples of 4, rows 8- (fp16) / 16-byte aligned", who);
    DCLIP_REQUIRE(out >= 0 && out <= 2 && (out != 2 || x_f16), "%s: output dtype 0 bf16 / 1 f32 (/ 2 f16 with f16 input)", who);
    const int nv = (int)((D + 255) / 256);
    const dim3 grid((unsigned)((M + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    TraceScope tr(DCLIP_TRACE_LAYERNORM, 8.0 * (double)M * D, (double)M * D * ((x_f16 ? 2.0 : 4.0) + (out == 1 ? 4.0 : 2.0)), stream);
#define LN_GO(OUTV, XH) hipLaunchKernelGGL((ln_fwd_kernel<NV, OUTV, XH>), grid, dim3(256), 0, st, x, ldx, row_index, gamma, beta, y, ldy, mean, rstd, (int)M, (int)D, eps)
    LN_DISPATCH(nv,
        if (x_f16) { if (out == 2) LN_GO(2, true); else if (out == 1) LN_GO(1, true); else LN_GO(0, true); }
        else { if (out == 1) LN_GO(1, false); else LN_GO(0, false); });
#undef LN_GO
    return dclip_check_launch(who);
}

extern "C" int dclip_layernorm_fwd(const float* x, int64_t ldx, const int32_t* row_index, const float* gamma,
                                   const float* beta, void* y, int64_t ldy, int out_f32, float* mean, float* rstd,
                                   int64_t M, int64_t D, float eps, void* stream) {
    return ln_fwd_launch(x, 0, ldx, row_index, gamma, beta, y, ldy, out_f32 ? 1 : 0, mean, rstd, M, D, eps, stream, "dclip_layernorm_fwd");
}

extern "C" int dclip_layernorm_fwd_f16(const void* x, int64_t ldx, const int32_t* row_index, const float* gamma,
                                       const float* beta, void* y, int64_t ldy, int out_dtype, float* mean, float* rstd,
                                       int64_t M, int64_t D, float eps, void* stream) {
    return ln_fwd_launch(x, 1, ldx, row_index, gamma, beta, y, ldy, out_dtype, mean, rstd, M, D, eps, stream, "dclip_layernorm_fwd_f16");
}

extern "C" int dclip_layernorm_bwd(const void* dy, int64_t lddy, int dy_f32, const float* x, int64_t ldx,
                                   const int32_t* row_index, const float* gamma, const float* mean, const float* rstd,
                                   float* dx_acc, int64_t lddx, void* dx_bf16, int64_t lddb, float* dgamma, float* dbeta,
                                   float* colsum_acc, int64_t M, int64_t D, void* stream) {
    DCLIP_REQUIRE(dy && x && gamma && mean && rstd && dx_acc, "dclip_layernorm_bwd: null operand");
    DCLIP_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 1024, "dclip_layernorm_bwd: need 0 < D <= 1024, D %% 4 == 0 (D=%ld)", (long)D);
    const int nv = (int)((D + 255) / 256);
    // persistent grid of 8-wave blocks, one per CU (the kernel's registers allow 2 waves per SIMD): every block ends with 3 x D
    // float atomics onto the same 3 D / 32 cache lines, so fewer, fatter blocks are cheaper — as long as every CU has one
    int blocks = (int)((M + 7) / 8);
    if (blocks > 256) blocks = 256;
    // algorithmic bytes: read dy (2 or 4), x (4), dx_acc (4) ; write dx_acc (4) + optional bf16 copy (2)
    TraceScope tr(DCLIP_TRACE_LN_BWD, 12.0 * (double)M * D, (double)M * D * ((dy_f32 ? 4.0 : 2.0) + 12.0 + (dx_bf16 ? 2.0 : 0.0)), stream, (int)M, (int)D, 0, 0);
    hipStream_t st = (hipStream_t)stream;
    LN_DISPATCH(nv,
        if (dy_f32) hipLaunchKernelGGL((ln_bwd_kernel<NV, true>), dim3(blocks), dim3(512), (size_t)3 * 8 * NV * 256 * 4, st, dy, lddy, x, ldx, row_index, gamma, mean, rstd, dx_acc, lddx, (bf16_t*)dx_bf16, lddb, dgamma, dbeta, colsum_acc, (int)M, (int)D);
        else hipLaunchKernelGGL((ln_bwd_kernel<NV, false>), dim3(blocks), dim3(512), (size_t)3 * 8 * NV * 256 * 4, st, dy, lddy, x, ldx, row_index, gamma, mean, rstd, dx_acc, lddx, (bf16_t*)dx_bf16, lddb, dgamma, dbeta, colsum_acc, (int)M, (int)D));
    return dclip_check_launch("dclip_layernorm_bwd");
}
