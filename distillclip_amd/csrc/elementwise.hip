// HBM-bound helper kernels around the GEMMs: dtype casts / weight transposes, im2row for the patch embedding,
// token-embedding gather / scatter-add, positional / class-token tables and their gradients, EOT pick indices.
#include "common.h"
#include <limits.h>

namespace {

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * 1024;
    for (; i < n; i += stride) {
        if (i + 3 < n) {
            const float4 v = *(const float4*)(s + i);
            bf16x4 o = {f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
            *(bf16x4*)(d + i) = o;
        } else {
            for (int64_t j = i; j < n; ++j) d[j] = f2bf(s[j]);
        }
    }
}

// f16 -> f32 (hidden states of the frozen teacher leave its fp16 residual stream as f32 tensors)
__global__ __launch_bounds__(256) void cast_f16_f32_kernel(const _Float16* __restrict__ s, float* __restrict__ d, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * 1024;
    for (; i < n; i += stride) {
        if (i + 3 < n) {
            const f16x4 v = *(const f16x4*)(s + i);
            *(float4*)(d + i) = float4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        } else {
            for (int64_t j = i; j < n; ++j) d[j] = (float)s[j];
        }
    }
}

// W f32 [R,C] -> Wb bf16 [R,C] and Wt bf16 [C,R]; 64x64 tiles through LDS
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ W, bf16_t* __restrict__ Wb,
                                                             bf16_t* __restrict__ Wt, int R, int C) {
    __shared__ bf16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        bf16_t v = f2bf(0.f);
        if (r < R && c < C) {
            v = f2bf(W[(int64_t)r * C + c]);
            if (Wb) Wb[(int64_t)r * C + c] = v;
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    if (Wt)
        for (int i = ty; i < 64; i += 4) {
            const int c = c0 + i, r = r0 + tx;
            if (r < R && c < C) Wt[(int64_t)c * R + r] = tile[tx][i];
        }
}

// Multi-tensor form of the above (the per-step refresh of a student's bf16 weight cache: one launch per tower instead of one
// per weight): up to CT_MAXJ jobs per launch, every R and C a multiple of 64; float4 reads, 8-byte bf16 stores both ways.
constexpr int CT_MAXJ = 24;
struct CastJobs {
    const float* W[CT_MAXJ];
    bf16_t* Wb[CT_MAXJ];
    bf16_t* Wt[CT_MAXJ];
    int R[CT_MAXJ], C[CT_MAXJ];
    int tile_end[CT_MAXJ];             // running count of 64 x 64 tiles
    int n;
};

__global__ __launch_bounds__(256) void cast_transpose_multi_kernel(CastJobs jb) {
    __shared__ __attribute__((aligned(8))) bf16_t tile[64][68];
    int j = 0;
    while (j + 1 < jb.n && (int)blockIdx.x >= jb.tile_end[j]) ++j;
    const int t = blockIdx.x - (j ? jb.tile_end[j - 1] : 0);
    const int R = jb.R[j], C = jb.C[j];
    const int tc = C >> 6;
    const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
    const float* __restrict__ W = jb.W[j];
    bf16_t* __restrict__ Wb = jb.Wb[j];
    bf16_t* __restrict__ Wt = jb.Wt[j];
    const int q = (threadIdx.x & 15) * 4, y = threadIdx.x >> 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = y + 16 * k;
        const float4 v = *(const float4*)(W + (int64_t)(r0 + r) * C + c0 + q);
        const bf16x4 o = {f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
        if (Wb) *(bf16x4*)(Wb + (int64_t)(r0 + r) * C + c0 + q) = o;
        *(bf16x4*)&tile[r][q] = o;
    }
    __syncthreads();
    if (Wt) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = y + 16 * k;
            const bf16x4 o = {tile[q][c], tile[q + 1][c], tile[q + 2][c], tile[q + 3][c]};
            *(bf16x4*)(Wt + (int64_t)(c0 + c) * R + r0 + q) = o;
        }
    }
}

// img f32 [B,C,res,res] -> rows bf16 [B*(G*G+cls), C*p*p]; row (b, cls + py*G + px), col (c, ky, kx)
// (reference _common.py:196-198 / timm PatchEmbed: Conv2d(k=p, s=p) == im2row + GEMM; the trailing res % p pixels
// are dropped exactly like the strided conv does).  cls rows are zero so the GEMM emits 0 there.
__global__ __launch_bounds__(256) void im2row_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int B, int C,
                                                     int res, int p, int G, int cls) {
    const int K = C * p * p;
    const int kq = K / 4;                       // float4 chunks per row
    const int rows = B * (G * G + cls);
    const int64_t total = (int64_t)rows * kq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int row = (int)(i / kq), q = (int)(i % kq);
        const int b = row / (G * G + cls), t = row % (G * G + cls);
        bf16x4 o = {f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
        if (t >= cls) {
            const int pi = t - cls, py = pi / G, px = pi % G;
            const int k = q * 4, c = k / (p * p), ky = (k / p) % p, kx = k % p;
            const float4 v = *(const float4*)(img + (((int64_t)b * C + c) * res + (py * p + ky)) * res + px * p + kx);
            o = bf16x4{f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
        }
        *(bf16x4*)(out + (int64_t)row * K + q * 4) = o;
    }
}

// out[0] = pos[0] + cls ; out[n>=1] = pos[n] + bias      (cls / bias may be null)
__global__ void token_table_kernel(const float* __restrict__ pos, const float* __restrict__ cls,
                                   const float* __restrict__ bias, float* __restrict__ out, int ntok, int D) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ntok * D) return;
    const int n = i / D, c = i % D;
    float v = pos[i];
    if (cls) v += (n == 0) ? cls[c] : (bias ? bias[c] : 0.f);
    else if (bias) v += bias[c];
    out[i] = v;
}

// tok_sum[n] = sum_b G[b,n,:]  ->  dpos += tok_sum ; dcls += tok_sum[0] ; dbias += sum_{n>=cls} tok_sum[n]
// grid (ceil(D / 256), ntok): one (token, column) per thread; the row-0 workgroups also reduce the bias column sums (loads only,
// so they pipeline) — the previous one-thread-per-column loop was a chain of 50-77 dependent read-modify-writes at the very end
// of a tower's backward
__global__ void token_table_bwd_kernel(const float* __restrict__ ts, float* __restrict__ dpos, float* __restrict__ dcls,
                                       float* __restrict__ dbias, int ntok, int D, int has_cls) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= D) return;
    const int n = blockIdx.y;
    const float v = ts[n * D + c];
    if (dpos) dpos[n * D + c] += v;
    if (n == 0) {
        if (dcls && has_cls) dcls[c] += v;
        if (dbias) {
            float sb = 0.f;
            for (int m = has_cls; m < ntok; ++m) sb += ts[m * D + c];
            dbias[c] += sb;
        }
    }
}

// out[n,:] += sum_b G[b,n,:]   grid (ceil(D/256), N, bsplit)
__global__ __launch_bounds__(256) void batch_sum_kernel(const float* __restrict__ G, float* __restrict__ out, int B, int N,
                                                        int D, int per) {
    // 16-byte loads, four samples in flight per trip (D % 4 == 0: checked by the caller)
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= D) return;
    const int n = blockIdx.y;
    const int b0 = blockIdx.z * per, b1 = min(B, b0 + per);
    const int64_t bs = (int64_t)N * D;
    const float* g = G + (int64_t)n * D + c;
    float4 s = {0.f, 0.f, 0.f, 0.f};
    int b = b0;
    for (; b + 4 <= b1; b += 4) {
        const float4 v0 = *(const float4*)(g + b * bs), v1 = *(const float4*)(g + (b + 1) * bs);
        const float4 v2 = *(const float4*)(g + (b + 2) * bs), v3 = *(const float4*)(g + (b + 3) * bs);
        s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
        s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; b < b1; ++b) {
        const float4 v = *(const float4*)(g + b * bs);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float* o = out + (int64_t)n * D + c;
    unsafeAtomicAdd(o, s.x); unsafeAtomicAdd(o + 1, s.y); unsafeAtomicAdd(o + 2, s.z); unsafeAtomicAdd(o + 3, s.w);
}

template <int OUT>      // 0 bf16, 1 f32, 2 f16
__global__ __launch_bounds__(256) void embed_gather_kernel(const int64_t* __restrict__ ids, int id_stride,
                                                           const float* __restrict__ table, const float* __restrict__ pos,
                                                           void* __restrict__ out, int rows, int N, int D) {
    const int dq = D / 4;
    const int64_t total = (int64_t)rows * dq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / dq), q = (int)(i % dq);
        const int64_t id = ids[(int64_t)(r / N) * id_stride + (r % N)];      // rows of N tokens out of id_stride per caption
        float4 v = *(const float4*)(table + id * D + q * 4);
        if (pos) {
            const float4 pv = *(const float4*)(pos + (int64_t)(r % N) * D + q * 4);
            v.x += pv.x; v.y += pv.y; v.z += pv.z; v.w += pv.w;
        }
        if (OUT == 1) *(float4*)((float*)out + (int64_t)r * D + q * 4) = v;
        else if (OUT == 2) *(f16x4*)((_Float16*)out + (int64_t)r * D + q * 4) = f16x4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        else *(bf16x4*)((bf16_t*)out + (int64_t)r * D + q * 4) = bf16x4{f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
    }
}

// Rows whose id is one of the three "hot" ids (padding 0, SOT, EOT in the clip.tokenize layout: every caption has them, so
// thousands of rows would hammer three table rows with contended atomics) are skipped here and reduced by the kernel below.
template <bool IN_F32>
__global__ __launch_bounds__(256) void embed_scatter_kernel(const int64_t* __restrict__ ids, const void* __restrict__ dx,
                                                            float* __restrict__ dtable, int rows, int D, int64_t h0, int64_t h1,
                                                            int64_t h2) {
    // one wave per token row: the id is read once, rows of the three hot ids (two thirds of a caption batch is padding) leave at
    // once (an element-per-thread grid paid a 64-bit division and an id load per element, also for the rows it then skipped)
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * 4;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += nw) {
        const int64_t id = ids[r];
        if (id == h0 || id == h1 || id == h2) continue;
        float* dst = dtable + id * D;
        // a lane per column, columns lane, lane + 64, ...: every atomic instruction covers 256 contiguous bytes (atomics retire per
        // instruction and per line touched: 16-byte loads with four strided atomics per lane took twice as long); loads first
        constexpr int CH = 4;
        for (int c0 = lane; c0 < D; c0 += 64 * CH) {
            float v[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int c = c0 + 64 * u;
                v[u] = c < D ? (IN_F32 ? ((const float*)dx)[(int64_t)r * D + c] : bf2f(((const bf16_t*)dx)[(int64_t)r * D + c])) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int c = c0 + 64 * u;
                if (c < D) unsafeAtomicAdd(dst + c, v[u]);
            }
        }
    }
}

// dtable[h_k] += sum over rows with ids == h_k of dx[row]   grid (ceil(D/256), row chunks)
template <bool IN_F32>
__global__ __launch_bounds__(256) void embed_hot_kernel(const int64_t* __restrict__ ids, const void* __restrict__ dx,
                                                        float* __restrict__ dtable, int rows, int D, int rows_per_block,
                                                        int64_t h0, int64_t h1, int64_t h2) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    bool s0 = false, s1 = false, s2 = false;
    const bool live = c < D;
    auto val = [&](int r) { return live ? (IN_F32 ? ((const float*)dx)[(int64_t)r * D + c] : bf2f(((const bf16_t*)dx)[(int64_t)r * D + c])) : 0.f; };
    // four rows per trip with unconditional loads (the ids are wave-uniform: the selects below cost nothing, the loads overlap)
    int r = r0;
    for (; r + 4 <= r1; r += 4) {
        int64_t id[4]; float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { id[u] = ids[r + u]; v[u] = val(r + u); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (id[u] == h0) { a0 += v[u]; s0 = true; }
            else if (id[u] == h1) { a1 += v[u]; s1 = true; }
            else if (id[u] == h2) { a2 += v[u]; s2 = true; }
        }
    }
    for (; r < r1; ++r) {
        const int64_t id = ids[r];
        if (id != h0 && id != h1 && id != h2) continue;
        const float v = val(r);
        if (id == h0) { a0 += v; s0 = true; }
        else if (id == h1) { a1 += v; s1 = true; }
        else { a2 += v; s2 = true; }
    }
    if (!live) return;
    if (s0) unsafeAtomicAdd(dtable + h0 * D + c, a0);
    if (s1) unsafeAtomicAdd(dtable + h1 * D + c, a1);
    if (s2) unsafeAtomicAdd(dtable + h2 * D + c, a2);
}

// idx[b] = b*N + argmax_n ids[b,n] (first maximum, like torch.argmax) ; ids == null -> b*N (class-token row).
// One wave per caption (a thread per caption walked its 77 ids as a dependent load chain: 19 us for 512 captions).
__global__ __launch_bounds__(256) void pick_index_kernel(const int64_t* __restrict__ ids, int id_stride, int* __restrict__ idx, int B, int N) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    int best = 0;
    if (ids) {
        // per-lane first maximum over n = lane, lane + 64, ... (argmax over the FULL caption, reference text_encoder.py:86), then the
        // wave's: larger value wins, equal values the smaller position
        long long m = LLONG_MIN;
        int pos = INT_MAX;
        for (int n = lane; n < id_stride; n += 64) {
            const long long v = ids[(int64_t)b * id_stride + n];
            if (v > m) { m = v; pos = n; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const long long om = __shfl_xor(m, o);
            const int op = __shfl_xor(pos, o);
            if (om > m || (om == m && op < pos)) { m = om; pos = op; }
        }
        best = pos < N ? pos : N - 1;                 // contract: the EOT lies inside the processed prefix
    }
    if (lane == 0) idx[b] = b * N + best;
}

// out[r, :] = src[idx[r], :]   (f32 rows)
__global__ void gather_rows_kernel(const float* __restrict__ src, int64_t lds, const int* __restrict__ idx,
                                   float* __restrict__ out, int rows, int D) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)rows * D) return;
    const int r = (int)(i / D), c = (int)(i % D);
    out[i] = src[(int64_t)idx[r] * lds + c];
}

// torch.optim.AdamW semantics (decoupled weight decay; bias-corrected), reference distil_model.py:160-162
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float lr, float b1, float b2,
                                                    float eps, float wd, float bc1, float bc2_sqrt, int zero_grad) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i];
        float pi = p[i] * (1.f - lr * wd);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= (lr / bc1) * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (zero_grad) g[i] = 0.f;                // the gradient is consumed: leave the accumulator clean for the next backward
    }
}

// float4 form of adamw_kernel (same per-element arithmetic).  The loads of the next grid-stride iteration are issued before the
// stores of the current one: vmcnt retires loads and stores in one in-order queue, so loads that follow stores wait for the
// stores' acknowledgements as well.
__device__ __forceinline__ void adamw_elem(float& pi, float gi, float& mi, float& vi, float lr, float b1, float b2, float eps,
                                           float wd, float bc1, float bc2_sqrt) {
    pi = pi * (1.f - lr * wd);
    mi = b1 * mi + (1.f - b1) * gi;
    vi = b2 * vi + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
}
__global__ __launch_bounds__(256) void adamw4_kernel(float4* __restrict__ p, float4* __restrict__ g, float4* __restrict__ m,
                                                     float4* __restrict__ v, int64_t n4, float lr, float b1, float b2, float eps,
                                                     float wd, float bc1, float bc2_sqrt, int zero_grad) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 pi = p[i], gi = g[i], mi = m[i], vi = v[i];
    for (;;) {
        const int64_t nx = i + stride;
        const bool more = nx < n4;
        float4 pn = pi, gn = gi, mn = mi, vn = vi;
        if (more) { pn = p[nx]; gn = g[nx]; mn = m[nx]; vn = v[nx]; }
        adamw_elem(pi.x, gi.x, mi.x, vi.x, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
        adamw_elem(pi.y, gi.y, mi.y, vi.y, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
        adamw_elem(pi.z, gi.z, mi.z, vi.z, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
        adamw_elem(pi.w, gi.w, mi.w, vi.w, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (zero_grad) g[i] = float4{0.f, 0.f, 0.f, 0.f};
        if (!more) break;
        i = nx; pi = pn; gi = gn; mi = mn; vi = vn;
    }
}

// dst += src (f32) ; optional bf16 copy of the updated dst.  Four independent 16-byte pieces per thread and trip: all their loads are in
// flight before the first store.
__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ dst, const float* __restrict__ src, bf16_t* __restrict__ dst_bf16, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 1024;
    for (int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i0 < n; i0 += 4 * stride) {
        float4 sv[4], dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < n) { sv[u] = *(const float4*)(src + i); dv[u] = *(const float4*)(dst + i); }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < n) {
                float4 d = dv[u];
                d.x += sv[u].x; d.y += sv[u].y; d.z += sv[u].z; d.w += sv[u].w;
                *(float4*)(dst + i) = d;
                if (dst_bf16) *(bf16x4*)(dst_bf16 + i) = bf16x4{f2bf(d.x), f2bf(d.y), f2bf(d.z), f2bf(d.w)};
            }
        }
    }
}

// The same with column sums of src (row length D: the bias gradient of the linear that wrote the residual stream).  A block owns 256 columns
// of a row range; thread (cg = tid & 63, rl = tid >> 6) walks rows r0 + rl, + 4, ... of its four columns with four rows in flight, the four
// row lanes are combined through LDS: ONE atomic per column and block.  (Round 5.  Until then: one atomic per ELEMENT — 13 M atomics onto the
// 16 cache lines of a 512-column sum made the hidden-state gradient add of a 25600 x 512 tensor a 3.2 ms launch — then four per thread of a
// grid-stride loop, 0.5 M per launch, ~100 us: the same lines, contended by every CU.)
__global__ __launch_bounds__(256) void axpy_colsum_kernel(float* __restrict__ dst, const float* __restrict__ src, bf16_t* __restrict__ dst_bf16,
                                                          int M, int D, float* __restrict__ colsum, int rows_per_block) {
    __shared__ float red[4][256];
    const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.x * 256 + cg * 4;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float4 part = float4{0.f, 0.f, 0.f, 0.f};
    if (col < D) {
        for (int r = r0 + rl; r < r1; r += 16) {
            float4 sv[4], dv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r + 4 * u < r1) {
                    const int64_t i = (int64_t)(r + 4 * u) * D + col;
                    sv[u] = *(const float4*)(src + i); dv[u] = *(const float4*)(dst + i);
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r + 4 * u < r1) {
                    const int64_t i = (int64_t)(r + 4 * u) * D + col;
                    float4 d = dv[u];
                    d.x += sv[u].x; d.y += sv[u].y; d.z += sv[u].z; d.w += sv[u].w;
                    *(float4*)(dst + i) = d;
                    if (dst_bf16) *(bf16x4*)(dst_bf16 + i) = bf16x4{f2bf(d.x), f2bf(d.y), f2bf(d.z), f2bf(d.w)};
                    part.x += sv[u].x; part.y += sv[u].y; part.z += sv[u].z; part.w += sv[u].w;
                }
        }
    }
    red[rl][cg * 4] = part.x; red[rl][cg * 4 + 1] = part.y; red[rl][cg * 4 + 2] = part.z; red[rl][cg * 4 + 3] = part.w;
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < D) unsafeAtomicAdd(colsum + c, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

inline int grid_for(int64_t work, int per_block = 256, int cap = 2048 * 4) {
    int64_t g = (work + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int dclip_cast_bf16(const float* src, void* dst, int64_t n, void* stream) {
    DCLIP_REQUIRE(src && dst && n > 0, "dclip_cast_bf16: bad argument");
    DCLIP_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 8) == 0, "dclip_cast_bf16: misaligned buffer");
    hipLaunchKernelGGL(cast_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
    return dclip_check_launch("dclip_cast_bf16");
}

extern "C" int dclip_cast_f16_f32(const void* src, float* dst, int64_t n, void* stream) {
    DCLIP_REQUIRE(src && dst && n > 0, "dclip_cast_f16_f32: bad argument");
    DCLIP_REQUIRE(((uintptr_t)src % 8) == 0 && ((uintptr_t)dst % 16) == 0, "dclip_cast_f16_f32: misaligned buffer");
    hipLaunchKernelGGL(cast_f16_f32_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)src, dst, n);
    return dclip_check_launch("dclip_cast_f16_f32");
}

extern "C" int dclip_cast_transpose_bf16(const float* W, void* Wb, void* Wt, int64_t R, int64_t C, void* stream) {
    DCLIP_REQUIRE(W && (Wb || Wt) && R > 0 && C > 0, "dclip_cast_transpose_bf16: bad argument");
    dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64));
    hipLaunchKernelGGL(cast_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, W, (bf16_t*)Wb, (bf16_t*)Wt, (int)R, (int)C);
    return dclip_check_launch("dclip_cast_transpose_bf16");
}

extern "C" int dclip_cast_transpose_bf16_multi(const float* const* W, void* const* Wb, void* const* Wt, const int64_t* R,
                                               const int64_t* C, int64_t n, void* stream) {
    DCLIP_REQUIRE(W && Wb && Wt && R && C && n > 0, "dclip_cast_transpose_bf16_multi: bad argument");
    CastJobs jb;
    jb.n = 0;
    int tiles = 0;
    auto flush = [&]() {
        if (jb.n) hipLaunchKernelGGL(cast_transpose_multi_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, jb);
        jb.n = 0; tiles = 0;
    };
    for (int64_t i = 0; i < n; ++i) {
        DCLIP_REQUIRE(W[i] && (Wb[i] || Wt[i]) && R[i] > 0 && C[i] > 0, "dclip_cast_transpose_bf16_multi: bad job %ld", (long)i);
        if (R[i] % 64 || C[i] % 64) {                    // odd shapes: the single-tensor kernel
            hipLaunchKernelGGL(cast_transpose_kernel, dim3((unsigned)((C[i] + 63) / 64), (unsigned)((R[i] + 63) / 64)), dim3(256), 0,
                               (hipStream_t)stream, W[i], (bf16_t*)Wb[i], (bf16_t*)Wt[i], (int)R[i], (int)C[i]);
            continue;
        }
        if (jb.n == CT_MAXJ) flush();
        const int k = jb.n++;
        jb.W[k] = W[i]; jb.Wb[k] = (bf16_t*)Wb[i]; jb.Wt[k] = (bf16_t*)Wt[i]; jb.R[k] = (int)R[i]; jb.C[k] = (int)C[i];
        tiles += (int)((R[i] / 64) * (C[i] / 64));
        jb.tile_end[k] = tiles;
    }
    flush();
    return dclip_check_launch("dclip_cast_transpose_bf16_multi");
}

extern "C" int dclip_im2row(const float* img, void* rows, int64_t B, int64_t C, int64_t res, int64_t patch, int cls_rows,
                            void* stream) {
    DCLIP_REQUIRE(img && rows && B > 0 && C > 0 && res >= patch && patch > 0, "dclip_im2row: bad argument");
    DCLIP_REQUIRE(patch % 4 == 0 && res % 4 == 0, "dclip_im2row: patch and resolution must be multiples of 4");
    DCLIP_REQUIRE(cls_rows == 0 || cls_rows == 1, "dclip_im2row: cls_rows must be 0 or 1");
    const int G = (int)(res / patch);
    const int64_t work = B * (G * G + cls_rows) * (C * patch * patch / 4);
    hipLaunchKernelGGL(im2row_kernel, dim3(grid_for(work)), dim3(256), 0, (hipStream_t)stream, img, (bf16_t*)rows, (int)B,
                       (int)C, (int)res, (int)patch, G, cls_rows);
    return dclip_check_launch("dclip_im2row");
}

extern "C" int dclip_token_table(const float* pos, const float* cls, const float* bias, float* out, int64_t ntok, int64_t D,
                                 void* stream) {
    DCLIP_REQUIRE(pos && out && ntok > 0 && D > 0, "dclip_token_table: bad argument");
    hipLaunchKernelGGL(token_table_kernel, dim3((unsigned)((ntok * D + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pos, cls,
                       bias, out, (int)ntok, (int)D);
    return dclip_check_launch("dclip_token_table");
}

extern "C" int dclip_token_table_bwd(const float* tok_sum, float* dpos, float* dcls, float* dbias, int64_t ntok, int64_t D,
                                     int has_cls, void* stream) {
    DCLIP_REQUIRE(tok_sum && ntok > 0 && D > 0, "dclip_token_table_bwd: bad argument");
    hipLaunchKernelGGL(token_table_bwd_kernel, dim3((unsigned)((D + 255) / 256), (unsigned)ntok), dim3(256), 0, (hipStream_t)stream, tok_sum,
                       dpos, dcls, dbias, (int)ntok, (int)D, has_cls);
    return dclip_check_launch("dclip_token_table_bwd");
}

extern "C" int dclip_batch_sum_acc(const float* G, float* out, int64_t B, int64_t N, int64_t D, void* stream) {
    DCLIP_REQUIRE(G && out && B > 0 && N > 0 && D > 0 && D % 4 == 0 && ((uintptr_t)G % 16) == 0, "dclip_batch_sum_acc: bad argument (D % 4 == 0, 16-byte aligned rows)");
    const int per = 32;
    dim3 grid((unsigned)((D / 4 + 255) / 256), (unsigned)N, (unsigned)((B + per - 1) / per));
    hipLaunchKernelGGL(batch_sum_kernel, grid, dim3(256), 0, (hipStream_t)stream, G, out, (int)B, (int)N, (int)D, per);
    return dclip_check_launch("dclip_batch_sum_acc");
}

extern "C" int dclip_embed_gather(const int64_t* ids, int64_t id_stride, const float* table, const float* pos, void* out,
                                  int out_dtype, int64_t rows, int64_t N, int64_t D, void* stream) {
    DCLIP_REQUIRE(ids && table && out && rows > 0 && N > 0 && id_stride >= N && D > 0 && D % 4 == 0, "dclip_embed_gather: bad argument");
    const dim3 grid(grid_for(rows * D / 4));
    DCLIP_REQUIRE(out_dtype >= 0 && out_dtype <= 2, "dclip_embed_gather: bad output dtype %d (0 bf16, 1 f32, 2 f16)", out_dtype);
    if (out_dtype == 1) hipLaunchKernelGGL((embed_gather_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, ids, (int)id_stride, table, pos, out, (int)rows, (int)N, (int)D);
    else if (out_dtype == 2) hipLaunchKernelGGL((embed_gather_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, ids, (int)id_stride, table, pos, out, (int)rows, (int)N, (int)D);
    else hipLaunchKernelGGL((embed_gather_kernel<0>), grid, dim3(256), 0, (hipStream_t)stream, ids, (int)id_stride, table, pos, out, (int)rows, (int)N, (int)D);
    return dclip_check_launch("dclip_embed_gather");
}

extern "C" int dclip_embed_scatter_add(const int64_t* ids, const void* dx, int dx_f32, float* dtable, int64_t rows, int64_t D,
                                       int64_t vocab, void* stream) {
    DCLIP_REQUIRE(ids && dx && dtable && rows > 0 && D > 0 && vocab >= 3, "dclip_embed_scatter_add: bad argument");
    const dim3 grid((unsigned)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096));
    const int64_t h0 = 0, h1 = vocab - 2, h2 = vocab - 1;       // padding, SOT, EOT (reference data/component/ms_coco.py:37)
    const int rpb = 128;
    const dim3 hgrid((unsigned)((D + 255) / 256), (unsigned)((rows + rpb - 1) / rpb));
    hipStream_t st = (hipStream_t)stream;
    if (dx_f32) {
        hipLaunchKernelGGL((embed_scatter_kernel<true>), grid, dim3(256), 0, st, ids, dx, dtable, (int)rows, (int)D, h0, h1, h2);
        hipLaunchKernelGGL((embed_hot_kernel<true>), hgrid, dim3(256), 0, st, ids, dx, dtable, (int)rows, (int)D, rpb, h0, h1, h2);
    } else {
        hipLaunchKernelGGL((embed_scatter_kernel<false>), grid, dim3(256), 0, st, ids, dx, dtable, (int)rows, (int)D, h0, h1, h2);
        hipLaunchKernelGGL((embed_hot_kernel<false>), hgrid, dim3(256), 0, st, ids, dx, dtable, (int)rows, (int)D, rpb, h0, h1, h2);
    }
    return dclip_check_launch("dclip_embed_scatter_add");
}

extern "C" int dclip_pick_index(const int64_t* ids, int64_t id_stride, int32_t* idx, int64_t B, int64_t N, void* stream) {
    DCLIP_REQUIRE(idx && B > 0 && N > 0 && id_stride >= N, "dclip_pick_index: bad argument");
    hipLaunchKernelGGL(pick_index_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ids, (int)id_stride, idx, (int)B, (int)N);
    return dclip_check_launch("dclip_pick_index");
}

extern "C" int dclip_gather_rows(const float* src, int64_t ld, const int32_t* idx, float* out, int64_t rows, int64_t D,
                                 void* stream) {
    DCLIP_REQUIRE(src && idx && out && rows > 0 && D > 0, "dclip_gather_rows: bad argument");
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((rows * D + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, ld, idx, out, (int)rows, (int)D);
    return dclip_check_launch("dclip_gather_rows");
}

// several ranges in one launch (blockIdx.y = range): the sharded data-parallel step updates one owned slice per gradient bucket — 9 launches
// of ~20 us each per step for the two l_clip students where two whole-tower launches do the same bytes
struct AdamwRanges { float4* p[DCLIP_ADAMW_MAX_RANGES]; float4* g[DCLIP_ADAMW_MAX_RANGES]; float4* m[DCLIP_ADAMW_MAX_RANGES]; float4* v[DCLIP_ADAMW_MAX_RANGES]; int64_t n4[DCLIP_ADAMW_MAX_RANGES]; };
__global__ __launch_bounds__(256) void adamw4_multi_kernel(AdamwRanges r, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                           int zero_grad) {
    const int k = blockIdx.y;
    float4* __restrict__ p = r.p[k]; float4* __restrict__ g = r.g[k]; float4* __restrict__ m = r.m[k]; float4* __restrict__ v = r.v[k];
    const int64_t n4 = r.n4[k];
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 pi = p[i], gi = g[i], mi = m[i], vi = v[i];
    for (;;) {
        const int64_t nx = i + stride;
        const bool more = nx < n4;
        float4 pn = pi, gn = gi, mn = mi, vn = vi;
        if (more) { pn = p[nx]; gn = g[nx]; mn = m[nx]; vn = v[nx]; }
        adamw_elem(pi.x, gi.x, mi.x, vi.x, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
        adamw_elem(pi.y, gi.y, mi.y, vi.y, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
        adamw_elem(pi.z, gi.z, mi.z, vi.z, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
        adamw_elem(pi.w, gi.w, mi.w, vi.w, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (zero_grad) g[i] = float4{0.f, 0.f, 0.f, 0.f};
        if (!more) break;
        i = nx; pi = pn; gi = gn; mi = mn; vi = vn;
    }
}

extern "C" int dclip_adamw_multi(float* const* p, float* const* g, float* const* m, float* const* v, const int64_t* n, int32_t count, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, int64_t step, int zero_grad, void* stream) {
    DCLIP_REQUIRE(p && g && m && v && n && count > 0 && count <= DCLIP_ADAMW_MAX_RANGES && step >= 1, "dclip_adamw_multi: bad argument (1..%d ranges)", DCLIP_ADAMW_MAX_RANGES);
    AdamwRanges r;
    int64_t longest = 0;
    for (int k = 0; k < count; ++k) {
        DCLIP_REQUIRE(p[k] && g[k] && m[k] && v[k] && n[k] > 0 && n[k] % 4 == 0 && ((((uintptr_t)p[k] | (uintptr_t)g[k] | (uintptr_t)m[k] | (uintptr_t)v[k]) & 15) == 0),
                      "dclip_adamw_multi: range %d must be non-empty, a multiple of 4 elements and 16-byte aligned", k);
        r.p[k] = (float4*)p[k]; r.g[k] = (float4*)g[k]; r.m[k] = (float4*)m[k]; r.v[k] = (float4*)v[k]; r.n4[k] = n[k] / 4;
        longest = r.n4[k] > longest ? r.n4[k] : longest;
    }
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = sqrtf(1.f - powf(beta2, (float)step));
    const int per_range = 8192 / count < 256 ? 256 : 8192 / count;          // (grid-stride loop: the longest range sets the width, capped)
    hipLaunchKernelGGL(adamw4_multi_kernel, dim3(grid_for(longest, 256, per_range), (unsigned)count), dim3(256), 0, (hipStream_t)stream, r, lr, beta1, beta2, eps,
                       weight_decay, bc1, bc2, zero_grad);
    return dclip_check_launch("dclip_adamw_multi");
}

extern "C" int dclip_adamw(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                           float eps, float weight_decay, int64_t step, int zero_grad, void* stream) {
    DCLIP_REQUIRE(p && g && m && v && n > 0 && step >= 1, "dclip_adamw: bad argument");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = sqrtf(1.f - powf(beta2, (float)step));
    // 16-byte aligned buffers (the flat parameter layout guarantees it; odd slices fall back): float4 kernel for the multiple-of-4
    // part, the scalar kernel for a tail of < 4 elements
    int64_t n4 = 0;
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) n4 = n / 4;
    if (n4 > 0)
        hipLaunchKernelGGL(adamw4_kernel, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream, (float4*)p, (float4*)g, (float4*)m,
                           (float4*)v, n4, lr, beta1, beta2, eps, weight_decay, bc1, bc2, zero_grad);
    const int64_t done = 4 * n4;
    if (done < n)
        hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n - done)), dim3(256), 0, (hipStream_t)stream, p + done, g + done, m + done,
                           v + done, n - done, lr, beta1, beta2, eps, weight_decay, bc1, bc2, zero_grad);
    return dclip_check_launch("dclip_adamw");
}

extern "C" int dclip_axpy_f32(float* dst, const float* src, void* dst_bf16, int64_t n, float* colsum_acc, int64_t D, void* stream) {
    DCLIP_REQUIRE(dst && src && n > 0 && n % 4 == 0 && D > 0 && D % 4 == 0, "dclip_axpy_f32: bad argument");
    if (!colsum_acc) {
        hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n, 4096, 2048)), dim3(256), 0, (hipStream_t)stream, dst, src, (bf16_t*)dst_bf16, n);
        return dclip_check_launch("dclip_axpy_f32");
    }
    DCLIP_REQUIRE(n % D == 0 && n / D < (1LL << 31), "dclip_axpy_f32: column sums need whole rows (n %% D == 0)");
    const int64_t M = n / D;
    const int colblocks = (int)((D + 255) / 256);
    // ~512 workgroups: enough rows in flight for the memory system, few enough atomics (256 per workgroup) onto the D sums
    int rows_per_block = (int)((M * colblocks + 511) / 512);
    rows_per_block = rows_per_block < 16 ? 16 : (rows_per_block + 15) & ~15;
    const dim3 grid((unsigned)colblocks, (unsigned)((M + rows_per_block - 1) / rows_per_block));
    hipLaunchKernelGGL(axpy_colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, dst, src, (bf16_t*)dst_bf16, (int)M, (int)D, colsum_acc, rows_per_block);
    return dclip_check_launch("dclip_axpy_f32");
}

