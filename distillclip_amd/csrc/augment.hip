// Training image transform after decode / resize / crop, on the GPU (gfx950):
//     RandAugment(num_ops) -> ToTensor -> Normalize        uint8 [B,H,W,3]  ->  f32 [B,3,H,W]
//
//   reference: data/component/ms_coco.py:15-26 (the transform chain), rand_augment.py:10-87 (_apply_op), :128-166 (op table
//   and the per-image loop), utils.py:11-12 (CLIP mean / std).  The reference runs the ops on PIL images (torchvision's
//   functional_pil path); this kernel reproduces Pillow's pixel arithmetic bit for bit (oracle/augment.py states each rule and
//   tests/test_augment_cpu.py pins the rules against the real Pillow):
//     geometric ops   Image.transform(AFFINE, NEAREST): 16.16 fixed-point source coordinates (Geometry.c affine_fixed);
//                     unit-scale translations take Pillow's ImagingScaleAffine path = an integer shift
//     Brightness / Contrast / Sharpness   ImageEnhance: Image.blend(degenerate, image, factor) in float32, truncating inside
//                     [0,1], clipping outside; degenerate = black / rounded mean luma (16.16 ITU-R 601) / 3x3 SMOOTH filter
//     Posterize       bit mask ;  AutoContrast / Equalize   ImageOps per-channel histogram look-up tables
// One workgroup per image walks the image's op list; stages ping-pong between two byte images in the workspace (150 KiB
// each at 224 px: they live in L2).  HBM-bound byte work: no MFMA, floating-point contraction is switched off for this file so
// that no fused multiply-add changes a rounding Pillow's scalar C code does not have.
#include <hip/hip_runtime.h>

#include "common.h"

// Pillow's scalar C code rounds every product and every sum: no fused multiply-add may be formed in this file (HIP's
// __fmul_rn / __fadd_rn are plain operators compiled under -ffp-contract=fast, so the file uses its own helpers).
#pragma clang fp contract(off)

namespace {

// individually rounded IEEE operations (defined after the pragma above, so that they carry no `contract` flag)
__device__ __forceinline__ float mul_r(float x, float y) { return x * y; }
__device__ __forceinline__ float add_r(float x, float y) { return x + y; }
__device__ __forceinline__ float sub_r(float x, float y) { return x - y; }
__device__ __forceinline__ float div_r(float x, float y) { return x / y; }
__device__ __forceinline__ double dmul_r(double x, double y) { return x * y; }
__device__ __forceinline__ double dadd_r(double x, double y) { return x + y; }

enum { OP_IDENTITY = 0, OP_AFFINE = 1, OP_SHIFT = 2, OP_BRIGHTNESS = 3, OP_CONTRAST = 4, OP_SHARPNESS = 5, OP_POSTERIZE = 6,
       OP_AUTOCONTRAST = 7, OP_EQUALIZE = 8, OP_COUNT = 9 };

struct AugArgs {
    const uint8_t* in; const dclip_aug_op* ops; int num_ops;
    int B, H, W;
    float mean[3], stdv[3];
    float* out; uint8_t* aug_out;
    uint8_t* ws;
};

__device__ __forceinline__ uint8_t blend_px(int d, int v, float alpha, bool inside) {
    const float t = add_r((float)d, mul_r(alpha, (float)(v - d)));
    if (inside) return (uint8_t)t;
    if (t <= 0.f) return 0;
    if (t >= 255.f) return 255;
    return (uint8_t)t;
}

constexpr int AUG_THREADS = 1024;

// dst = table[channel][src], pixel by pixel (three contiguous bytes per lane, no index division)
__device__ __forceinline__ void apply_lut(const uint8_t* __restrict__ rs, uint8_t* __restrict__ wd, const uint8_t (*lut)[256],
                                          int npix, int tid) {
#pragma unroll 4
    for (int p = tid; p < npix; p += AUG_THREADS) {
        const uint8_t* q = rs + p * 3;
        const uint8_t r = lut[0][q[0]], g = lut[1][q[1]], b = lut[2][q[2]];
        wd[p * 3 + 0] = r; wd[p * 3 + 1] = g; wd[p * 3 + 2] = b;
    }
}
      // one workgroup per image: the stages are latency-bound byte loops, so use every wave slot

__global__ __launch_bounds__(AUG_THREADS) void augment_kernel(AugArgs a) {
    __shared__ unsigned hist[3][256];
    __shared__ uint8_t lut[3][256];
    __shared__ float ntab[3][256];
    __shared__ unsigned long long red[AUG_THREADS / 64];
    const int tid = threadIdx.x;
    const int img = blockIdx.x;
    const int H = a.H, W = a.W, npix = H * W, nbytes = npix * 3;
    const uint8_t* src = a.in + (int64_t)img * nbytes;
    uint8_t* buf0 = a.ws + (int64_t)img * 2 * nbytes;
    uint8_t* buf1 = buf0 + nbytes;
    uint8_t* dst = buf0;
    for (int s = 0; s < a.num_ops; ++s) {
        const dclip_aug_op op = a.ops[(int64_t)img * a.num_ops + s];
        if (op.op == OP_IDENTITY) continue;
        // a stage reads one byte image and writes the other: restrict views let the loads of later iterations start before the
        // stores of earlier ones (otherwise every iteration is a full load -> store round trip)
        const uint8_t* __restrict__ rs = src;
        uint8_t* __restrict__ wd = dst;
        if (op.op == OP_AFFINE || op.op == OP_SHIFT) {
#pragma unroll 4
            for (int p = tid; p < npix; p += AUG_THREADS) {
                const int y = p / W, x = p - y * W;
                int xin, yin;
                if (op.op == OP_AFFINE) {
                    xin = (op.c[2] + op.c[1] * y + op.c[0] * x) >> 16;
                    yin = (op.c[5] + op.c[4] * y + op.c[3] * x) >> 16;
                } else {
                    xin = x + op.c[0];
                    yin = y + op.c[1];
                }
                uint8_t r = 0, g = 0, b = 0;
                if (xin >= 0 && xin < W && yin >= 0 && yin < H) {
                    const uint8_t* q = rs + (yin * W + xin) * 3;
                    r = q[0]; g = q[1]; b = q[2];
                }
                wd[p * 3 + 0] = r; wd[p * 3 + 1] = g; wd[p * 3 + 2] = b;
            }
        } else if (op.op == OP_BRIGHTNESS || op.op == OP_CONTRAST) {
            int deg = 0;
            if (op.op == OP_CONTRAST) {
                unsigned long long sum = 0;
                for (int p = tid; p < npix; p += AUG_THREADS) {
                    const uint8_t* q = rs + p * 3;
                    sum += (unsigned)((q[0] * 19595 + q[1] * 38470 + q[2] * 7471 + 0x8000) >> 16);
                }
                for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
                __syncthreads();
                if ((tid & 63) == 0) red[tid >> 6] = sum;
                __syncthreads();
                sum = 0;
                for (int w = 0; w < AUG_THREADS / 64; ++w) sum += red[w];
                deg = (int)((double)sum / (double)npix + 0.5);       // int(ImageStat.Stat(L).mean[0] + 0.5)
            }
            // the blend against a constant is a function of the byte alone: 256-entry table, then the shared table pass
            const bool inside = op.f >= 0.f && op.f <= 1.f;
            __syncthreads();
            if (tid < 256) lut[0][tid] = lut[1][tid] = lut[2][tid] = blend_px(deg, tid, op.f, inside);
            __syncthreads();
            apply_lut(rs, wd, lut, npix, tid);
        } else if (op.op == OP_SHARPNESS) {
            const float k1 = 1.f / 13.f, k5 = 5.f / 13.f;               // (FLOAT32) kernel[i] / divisor, as Filter.c stores them
            const bool inside = op.f >= 0.f && op.f <= 1.f;
#pragma unroll 4
            for (int i = tid; i < nbytes; i += AUG_THREADS) {
                const int p = i / 3, ch = i - p * 3;
                const int y = p / W, x = p - y * W;
                const int v = rs[i];
                int sm = v;                                              // one-pixel border: copied
                if (x > 0 && x < W - 1 && y > 0 && y < H - 1) {
                    float ss = 0.5f;
#pragma unroll
                    for (int dy = 1; dy >= -1; --dy) {
                        const uint8_t* q = rs + ((y + dy) * W + x) * 3 + ch;
                        const float kc = dy == 0 ? k5 : k1;
                        const float row = add_r(add_r(mul_r((float)q[-3], k1), mul_r((float)q[0], kc)),
                                                    mul_r((float)q[3], k1));
                        ss = add_r(ss, row);
                    }
                    const int t = (int)ss;
                    sm = t < 0 ? 0 : (t > 255 ? 255 : t);
                }
                wd[i] = blend_px(sm, v, op.f, inside);
            }
        } else if (op.op == OP_POSTERIZE) {
            const uint8_t mask = (uint8_t)op.c[0];
            __syncthreads();
            if (tid < 256) lut[0][tid] = lut[1][tid] = lut[2][tid] = (uint8_t)tid & mask;
            __syncthreads();
            apply_lut(rs, wd, lut, npix, tid);
        } else {                                                         // OP_AUTOCONTRAST / OP_EQUALIZE
            __syncthreads();
            for (int i = tid; i < 768; i += AUG_THREADS) (&hist[0][0])[i] = 0u;
            __syncthreads();
            for (int p = tid; p < npix; p += AUG_THREADS) {
                const uint8_t* q = rs + p * 3;
                atomicAdd(&hist[0][q[0]], 1u); atomicAdd(&hist[1][q[1]], 1u); atomicAdd(&hist[2][q[2]], 1u);
            }
            __syncthreads();
            if (op.op == OP_AUTOCONTRAST) {
                for (int ch = 0; ch < 3 && tid < 256; ++ch) {
                    int lo = 0, hi = 255;
                    while (lo < 256 && hist[ch][lo] == 0u) ++lo;
                    while (hi >= 0 && hist[ch][hi] == 0u) --hi;
                    int v = tid;
                    if (hi > lo) {
                        const double scale = 255.0 / (double)(hi - lo);
                        const double offset = -(double)lo * scale;
                        const double t = dadd_r(dmul_r((double)tid, scale), offset);
                        v = (int)t;
                        v = v < 0 ? 0 : (v > 255 ? 255 : v);
                    }
                    lut[ch][tid] = (uint8_t)v;
                }
            } else {
                if (tid < 3) {
                    const int ch = tid;
                    unsigned total = 0, last = 0;
                    int nz = 0;
                    for (int i = 0; i < 256; ++i)
                        if (hist[ch][i]) { total += hist[ch][i]; last = hist[ch][i]; ++nz; }
                    const unsigned step = nz <= 1 ? 0u : (total - last) / 255u;
                    unsigned n = step / 2u;
                    for (int i = 0; i < 256; ++i) {
                        const unsigned v = step ? n / step : (unsigned)i;
                        lut[ch][i] = (uint8_t)(v > 255u ? 255u : v);
                        n += hist[ch][i];
                    }
                }
            }
            __syncthreads();
            apply_lut(rs, wd, lut, npix, tid);
        }
        __threadfence_block();
        __syncthreads();                                                // the next stage reads what other lanes wrote
        src = dst;
        dst = dst == buf0 ? buf1 : buf0;
    }
    // ToTensor + Normalize: float32(v) / 255, - mean, / std (IEEE divisions, as torch's float32 kernels) — a function of
    // (channel, byte): 768 table entries instead of two divisions per element
    __syncthreads();
    if (tid < 768) {
        const int ch = tid >> 8, v = tid & 255;
        ntab[ch][v] = div_r(sub_r(div_r((float)v, 255.f), a.mean[ch]), a.stdv[ch]);
    }
    __syncthreads();
    float* __restrict__ o = a.out + (int64_t)img * nbytes;
    const uint8_t* __restrict__ fin = src;
#pragma unroll 4
    for (int p = tid; p < npix; p += AUG_THREADS) {
        const uint8_t* q = fin + p * 3;
        o[p] = ntab[0][q[0]]; o[npix + p] = ntab[1][q[1]]; o[2 * npix + p] = ntab[2][q[2]];
    }
    if (a.aug_out) {
        uint8_t* ao = a.aug_out + (int64_t)img * nbytes;
        for (int i = tid; i < nbytes; i += AUG_THREADS) ao[i] = src[i];
    }
}

}  // namespace

extern "C" size_t dclip_augment_workspace(int64_t B, int64_t H, int64_t W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * 2 * (size_t)H * (size_t)W * 3;
}

extern "C" int dclip_augment_normalize(const uint8_t* images, int64_t B, int64_t H, int64_t W, const dclip_aug_op* ops,
                                       int num_ops, const float* mean3, const float* std3, float* out, uint8_t* aug_out,
                                       void* workspace, size_t ws_bytes, void* stream) {
    DCLIP_REQUIRE(images && out && mean3 && std3, "dclip_augment_normalize: null operand");
    DCLIP_REQUIRE(B > 0 && H >= 3 && W >= 3 && H <= 4096 && W <= 4096, "dclip_augment_normalize: need B > 0 and 3 <= H, W <= 4096 (B=%ld H=%ld W=%ld)",
                  (long)B, (long)H, (long)W);
    DCLIP_REQUIRE(num_ops >= 0 && num_ops <= 16 && (num_ops == 0 || ops), "dclip_augment_normalize: 0 <= num_ops <= 16 with an op table");
    DCLIP_REQUIRE(num_ops == 0 || (workspace && ws_bytes >= dclip_augment_workspace(B, H, W)),
                  "dclip_augment_normalize: workspace too small (%zu < %zu)", ws_bytes, dclip_augment_workspace(B, H, W));
    DCLIP_REQUIRE(B < (1LL << 24), "dclip_augment_normalize: batch too large");
    AugArgs a;
    a.in = images; a.ops = ops; a.num_ops = num_ops; a.B = (int)B; a.H = (int)H; a.W = (int)W;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean3[c]; a.stdv[c] = std3[c]; }
    a.out = out; a.aug_out = aug_out; a.ws = (uint8_t*)workspace;
    hipLaunchKernelGGL(augment_kernel, dim3((unsigned)B), dim3(AUG_THREADS), 0, (hipStream_t)stream, a);
    return dclip_check_launch("dclip_augment_normalize");
}

// ------------------------------------------------------------------------------------------------------------------
// Resize(S) + CenterCrop(S) of decoded images of mixed sizes: Pillow's two-pass antialiased bilinear resample
// (libImaging Resample.c: 8-bit pass results, 22-bit fixed-point coefficients), evaluated only on the crop window.
//   reference: data/component/ms_coco.py:16-17,23-24 (transforms.Resize(224), transforms.CenterCrop(224) on PIL images)
// The coefficient tables depend on the source size only; the host builds them in double precision exactly as
// precompute_coeffs / normalize_coeffs_8bpc do (distillclip_amd/augment.py caches them per size).
// ------------------------------------------------------------------------------------------------------------------
namespace {

struct ResizeArgs {
    const uint8_t* packed; const dclip_resize_desc* desc; const int32_t* tables;
    int S;
    uint8_t* out; uint8_t* ws;
};

__device__ __forceinline__ uint8_t clip8_fixed(int v) {
    v >>= 22;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

__global__ __launch_bounds__(256) void resize_crop_kernel(ResizeArgs a) {
    const dclip_resize_desc d = a.desc[blockIdx.x];
    const int S = a.S, H = d.height, W = d.width;
    const uint8_t* src = a.packed + d.src_offset;
    const int32_t* hb = a.tables + d.table_offset;
    const int32_t* hk = hb + 2 * S;
    const int32_t* vb = hk + (int64_t)S * d.ksize_h;
    const int32_t* vk = vb + 2 * S;
    uint8_t* temp = a.ws + d.temp_offset;
    // horizontal pass over the source rows the crop window needs
    const int n1 = d.nrows * S * 3;
    for (int i = threadIdx.x; i < n1; i += 256) {
        const int ch = i % 3, t = i / 3;
        const int ox = t % S, r = t / S;
        const int row = min(d.row0 + r, H - 1);
        const int x0 = hb[ox * 2], n = hb[ox * 2 + 1];
        const int32_t* k = hk + (int64_t)ox * d.ksize_h;
        const uint8_t* p = src + ((int64_t)row * W) * 3 + ch;
        int ss = 1 << 21;
        for (int x = 0; x < n; ++x) ss += (int)p[min(x0 + x, W - 1) * 3] * k[x];
        temp[i] = clip8_fixed(ss);
    }
    __threadfence_block();
    __syncthreads();
    // vertical pass
    uint8_t* o = a.out + (int64_t)blockIdx.x * S * S * 3;
    const int n2 = S * S * 3;
    for (int i = threadIdx.x; i < n2; i += 256) {
        const int ch = i % 3, t = i / 3;
        const int ox = t % S, oy = t / S;
        const int y0 = vb[oy * 2], n = vb[oy * 2 + 1];
        const int32_t* k = vk + (int64_t)oy * d.ksize_v;
        int ss = 1 << 21;
        for (int y = 0; y < n; ++y) ss += (int)temp[(min(y0 + y, d.nrows - 1) * S + ox) * 3 + ch] * k[y];
        o[i] = clip8_fixed(ss);
    }
}

}  // namespace

extern "C" int dclip_resize_center_crop(const uint8_t* packed, const dclip_resize_desc* desc, const int32_t* tables, int64_t B,
                                        int64_t S, uint8_t* out, void* workspace, size_t ws_bytes, void* stream) {
    DCLIP_REQUIRE(packed && desc && tables && out && workspace, "dclip_resize_center_crop: null operand");
    DCLIP_REQUIRE(B > 0 && B < (1LL << 24) && S >= 1 && S <= 4096, "dclip_resize_center_crop: need B > 0, 1 <= S <= 4096 (B=%ld S=%ld)",
                  (long)B, (long)S);
    DCLIP_REQUIRE(ws_bytes > 0, "dclip_resize_center_crop: empty workspace");
    ResizeArgs a;
    a.packed = packed; a.desc = desc; a.tables = tables; a.S = (int)S; a.out = out; a.ws = (uint8_t*)workspace;
    hipLaunchKernelGGL(resize_crop_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, a);
    return dclip_check_launch("dclip_resize_center_crop");
}
