// Validation retrieval metrics (gfx950): top-k accuracy of the matching caption among all captions and the diagonal scores.
//
//   reference: model/dual_distill_model.py:271-275 (norm_and_logits: rows / |rows|, logits = img @ txt.T),
//              :204-212 (log_diag_score: mean softmax(logits, 1) diagonal, mean diagonal),
//              :220-224 (log_acc: torchmetrics multiclass accuracy(top_k = k) against labels arange(n)), k_list :87
//              model/distil_model.py:171-191, :224-231 (same metrics for the one-tower models)
//
// The [n, n] logits are never written: a workgroup owns 16 rows, recomputes 16 x 16 tiles on the exact-f32 MFMA
// (v_mfma_f32_16x16x4_f32, the same tile routine as the loss) and keeps, per row, the number of columns that beat the
// diagonal and the sum of exp(logit - 1) (cosine logits are bounded by 1).  rank < k  <=>  the label is inside the top k.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

constexpr int MAXK = 8;

struct MetricArgs {
    const float* img; const float* txt;       // raw embeddings [n, E]
    float* nimg; float* ntxt;                 // normalised copies (workspace)
    int* rank; float* sexp; float* diag;      // per-row results (workspace)
    int n, E, nk;
    int ks[MAXK];
    float* out;                               // [nk + 2]
    int* rank_out;                            // optional copy of the ranks
};

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// one wave per row: x / |x|  (no epsilon, as the reference)
__global__ __launch_bounds__(256) void metric_norm_kernel(MetricArgs a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= 2 * a.n) return;
    const float* src = row < a.n ? a.img + (int64_t)row * a.E : a.txt + (int64_t)(row - a.n) * a.E;
    float* dst = row < a.n ? a.nimg + (int64_t)row * a.E : a.ntxt + (int64_t)(row - a.n) * a.E;
    float s = 0.f;
    for (int c = lane * 4; c < a.E; c += 256) {
        const float4 v = *(const float4*)(src + c);
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    const float inv = 1.f / sqrtf(wsum(s));
    for (int c = lane * 4; c < a.E; c += 256) {
        const float4 v = *(const float4*)(src + c);
        *(float4*)(dst + c) = float4{v.x * inv, v.y * inv, v.z * inv, v.w * inv};
    }
}

// lane (r = l & 15, g = l >> 4) covers k = 16 o + 4 g + t at MFMA step t ; acc[q] = X[row 4 g + q][col l & 15]
__device__ __forceinline__ f32x4 cos_tile(const float* __restrict__ xa, const float* __restrict__ xb, int E, int lane) {
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

// grid = ceil(n / 16) ; the four waves split the column tiles
__global__ __launch_bounds__(256) void metric_stripe_kernel(MetricArgs a) {
    __shared__ float red_e[4][16];
    __shared__ int red_c[4][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = a.n, E = a.E;
    const int i0 = blockIdx.x * 16;
    const int ia = min(i0 + (lane & 15), n - 1) - (lane & 15);       // clamp the row panel inside the matrix
    const float* rows = a.nimg + (int64_t)ia * E;
    // the diagonal tile first: every later tile is compared with values produced by the very same instruction sequence
    float d[4];
    {
        const f32x4 S = cos_tile(rows, a.ntxt + (int64_t)ia * E, E, lane);
        const int g = lane >> 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) d[q] = __shfl(S[q], g * 16 + g * 4 + q);
    }
    int cnt[4] = {0, 0, 0, 0};
    float se[4] = {0.f, 0.f, 0.f, 0.f};
    const int ntile = (n + 15) / 16;
    for (int jt = wave; jt < ntile; jt += 4) {
        const int j0 = jt * 16;
        const int jb = min(j0 + (lane & 15), n - 1) - (lane & 15);
        const f32x4 S = cos_tile(rows, a.ntxt + (int64_t)jb * E, E, lane);
        const int col = j0 + (lane & 15);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = i0 + (lane >> 4) * 4 + q;
            if (col < n) {
                se[q] += __expf(S[q] - 1.f);
                cnt[q] += (col != row && S[q] > d[q]) ? 1 : 0;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float e = se[q];
        int c = cnt[q];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { e += __shfl_xor(e, o); c += __shfl_xor(c, o); }
        if ((lane & 15) == 0) { red_e[wave][(lane >> 4) * 4 + q] = e; red_c[wave][(lane >> 4) * 4 + q] = c; }
    }
    __syncthreads();
    if (threadIdx.x < 16 && i0 + threadIdx.x < n) {
        const int r = threadIdx.x;
        const float e = (red_e[0][r] + red_e[1][r]) + (red_e[2][r] + red_e[3][r]);
        const int c = (red_c[0][r] + red_c[1][r]) + (red_c[2][r] + red_c[3][r]);
        a.rank[i0 + r] = c;
        a.sexp[i0 + r] = e;                   // sum_j exp(logit_ij - 1); the finalize step forms exp(d_i - 1) / sum
    }
    // every wave holds the same d; wave 0 writes it (lane (g, 0) owns rows 4 g + q)
    if (wave == 0 && (lane & 15) == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = i0 + (lane >> 4) * 4 + q;
            if (row < n) a.diag[row] = d[q];
        }
    }
}

// one workgroup: deterministic reduction of the per-row results
__global__ __launch_bounds__(256) void metric_finalize_kernel(MetricArgs a) {
    __shared__ float red[4][MAXK + 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float hit[MAXK], sm = 0.f, dg = 0.f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) hit[k] = 0.f;
    for (int i = threadIdx.x; i < a.n; i += 256) {
        const int r = a.rank[i];
        const float d = a.diag[i];
#pragma unroll
        for (int k = 0; k < MAXK; ++k) hit[k] += (k < a.nk && r < a.ks[k]) ? 1.f : 0.f;
        sm += __expf(d - 1.f) / a.sexp[i];
        dg += d;
        if (a.rank_out) a.rank_out[i] = r;
    }
#pragma unroll
    for (int k = 0; k < MAXK; ++k) hit[k] = wsum(hit[k]);
    sm = wsum(sm); dg = wsum(dg);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < MAXK; ++k) red[wave][k] = hit[k];
        red[wave][MAXK] = sm; red[wave][MAXK + 1] = dg;
    }
    __syncthreads();
    if (threadIdx.x < MAXK + 2) {
        const int k = threadIdx.x;
        const float v = ((red[0][k] + red[1][k]) + (red[2][k] + red[3][k])) / (float)a.n;
        if (k < a.nk) a.out[k] = v;
        else if (k == MAXK) a.out[a.nk] = v;
        else if (k == MAXK + 1) a.out[a.nk + 1] = v;
    }
}

inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t dclip_retrieval_metrics_workspace(int64_t n, int64_t E) {
    if (n <= 0 || E <= 0) return 0;
    return 2 * up256((size_t)n * E * 4) + 3 * up256((size_t)n * 4);
}

extern "C" int dclip_retrieval_metrics(const float* img, const float* txt, int64_t n, int64_t E, const int32_t* ks, int nk,
                                       float* out, int32_t* rank_out, void* ws, size_t ws_bytes, void* stream) {
    DCLIP_REQUIRE(img && txt && out && ws, "dclip_retrieval_metrics: null operand");
    DCLIP_REQUIRE(n > 0 && n < (1LL << 30), "dclip_retrieval_metrics: need 0 < n < 2^30 (n=%ld)", (long)n);
    DCLIP_REQUIRE(E > 0 && E % 16 == 0, "dclip_retrieval_metrics: E=%ld must be a positive multiple of 16", (long)E);
    DCLIP_REQUIRE(nk >= 0 && nk <= MAXK && (nk == 0 || ks), "dclip_retrieval_metrics: 0 <= nk <= %d cut-offs (nk=%d)", MAXK, nk);
    DCLIP_REQUIRE(ws_bytes >= dclip_retrieval_metrics_workspace(n, E), "dclip_retrieval_metrics: workspace too small (%zu < %zu)",
                  ws_bytes, dclip_retrieval_metrics_workspace(n, E));
    DCLIP_REQUIRE(((uintptr_t)img % 16) == 0 && ((uintptr_t)txt % 16) == 0 && ((uintptr_t)ws % 16) == 0,
                  "dclip_retrieval_metrics: operands must be 16-byte aligned");
    MetricArgs a;
    a.img = img; a.txt = txt; a.n = (int)n; a.E = (int)E; a.nk = nk;
    for (int k = 0; k < MAXK; ++k) a.ks[k] = k < nk ? ks[k] : 0;
    char* p = (char*)ws;
    a.nimg = (float*)p; p += up256((size_t)n * E * 4);
    a.ntxt = (float*)p; p += up256((size_t)n * E * 4);
    a.rank = (int*)p; p += up256((size_t)n * 4);
    a.sexp = (float*)p; p += up256((size_t)n * 4);
    a.diag = (float*)p;
    a.out = out; a.rank_out = rank_out;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(metric_norm_kernel, dim3((unsigned)((2 * n + 3) / 4)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(metric_stripe_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(metric_finalize_kernel, dim3(1), dim3(256), 0, st, a);
    return dclip_check_launch("dclip_retrieval_metrics");
}
