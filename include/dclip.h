/*
 * dclip.h — C ABI of libdistillclip_hip.so (MI355X / gfx950).
 *
 * The reference (ForJadeForest/DistillCLIP) is 100 % Python on top of PyTorch ATen; it has no FFI of its own.
 * The boundary this library replaces is therefore the implicit ATen kernel sequence behind the reference's
 * nn.Module calls.  Each entry point cites the reference call site (file:line under the reference root) whose
 * arithmetic it implements.  All pointers are raw device pointers; `stream` is a hipStream_t passed as void*.
 * No synchronisation inside any entry point, no allocation, no global mutable state on the compute path: the process-global state is
 * the opt-in profiling hooks (dclip_trace_*: launch trace, GEMM stamps / clock stamps; the wgrad fallback counter) and a handful of
 * tuning knobs read from DCLIP_* environment variables, latched once on first use and constant
 * afterwards.  A dclip_encoder handle additionally remembers which workspace its last training forward
 * prepared for a backward (dclip_encoder_backward below).  Every function returns
 * 0 on success, DCLIP_EINVAL (-1) for a bad argument, DCLIP_ELAUNCH (-2) for a HIP launch failure, and
 * dclip_last_error_string() (thread-local) explains the last failure.
 *
 * dtypes: "bf16" = bfloat16 storage (MFMA operands), "f32" = float, "f16" = IEEE half (the frozen teacher's residual stream, the
 * 16-bit type the reference's `precision: 16` autocast keeps it in).  Row-major everywhere.
 *
 * Load order: the library links the HIP runtime (libamdhip64) the usual way.  A process that also loads PyTorch-ROCm must import torch
 * BEFORE this library is mapped, so that both resolve to the one HIP runtime torch ships (two runtimes in one process do not see each
 * other's device state: "no ROCm-capable device is detected").  dclip_runtime_check() reports that condition in words.
 */
#ifndef DCLIP_H
#define DCLIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DCLIP_OK 0
#define DCLIP_EINVAL (-1)
#define DCLIP_ELAUNCH (-2)

int dclip_version(void);                    /* ABI version, bumped on any signature change */
/* 0 when the HIP runtime this library is bound to sees a device; DCLIP_ELAUNCH otherwise, with dclip_last_error_string() naming the
 * likely cause (a second HIP runtime mapped before this one: see "Load order" above) */
int dclip_runtime_check(void);
const char* dclip_arch(void);               /* "gfx950" */
const char* dclip_last_error_string(void);  /* thread-local */

/* activation codes for dclip_gemm_nt */
#define DCLIP_ACT_NONE 0
#define DCLIP_ACT_QUICKGELU 1 /* reference model/component/_common.py:23-25 */
#define DCLIP_ACT_GELU 2      /* exact erf GELU: timm Mlp act, reference weight_share_model.py:177 */
#define DCLIP_ACT_DGELU 3     /* multiply by gelu'(aux_in): backward of DCLIP_ACT_GELU from the saved pre-activation */
#define DCLIP_ACT_MULAUX 4    /* multiply by the 8-bit gelu' in aux_in: backward of DCLIP_ACT_GELU_SAVE */
#define DCLIP_ACT_GELU_SAVE 5 /* DCLIP_ACT_GELU whose aux_out receives gelu'(pre-activation) as 8-bit fixed point (below) */
#define DCLIP_ACT_QUICKGELU_SAVE 6 /* DCLIP_ACT_QUICKGELU with its derivative saved the same way: a CLIP tower that trains (plain
                                      ImageEncoder / TextEncoder as student, reference image_encoder.py:23-25, text_encoder.py:45-47) */
/* output dtype codes of dclip_gemm_nt / dclip_embed_gather / dclip_layernorm_fwd_f16 */
#define DCLIP_OUT_BF16 0
#define DCLIP_OUT_F32 1
#define DCLIP_OUT_F16 2

/*
 * C[M,N] = epilogue(alpha * A[M,K] · B[N,K]^T)        (nn.Linear / F.linear / x @ proj / conv-as-GEMM)
 *   reference: _common.py:59,90,104-108,213 ; text_encoder.py:72 ; weight_share_model.py:90,132,177,364
 *   A, B bf16 (lda, ldb in elements; K % 64 == 0; 16-byte aligned rows).
 *   epilogue, in order: + bias[N] (f32, may be NULL) ; if aux_out: store pre-activation as bf16 [M,N] (ld = ldc) ;
 *   activation `act` (DCLIP_ACT_DGELU multiplies by gelu'(aux_in[M,N] bf16, ld = ldc); DCLIP_ACT_GELU_SAVE stores
 *   gelu'(pre-activation) in aux_out instead — ONE BYTE per element, uint8 [M,N] with ld = ldc bytes: code q = rint((g' + 0.13) * 255 / 1.26),
 *   value -0.13 + q * 1.26 / 255 (the derivative of the exact GELU lies in [-0.129, 1.129]; |error| <= 2.5e-3) — which
 *   DCLIP_ACT_MULAUX multiplies by: the training towers use that pair, the derivative being a few extra instructions in the
 *   forward epilogue and a single multiply in the backward one; DCLIP_ACT_QUICKGELU_SAVE is the same pair for QuickGELU, whose
 *   derivative s + 1.702 x s (1 - s), s = sigmoid(1.702 x), lies in [-0.1, 1.1]) ;
 *   + residual[M,N] (ld = ldr, may be NULL; f32 with bf16 / f32 output — may alias C when the output is f32 —, f16 with f16 output — may
 *   alias C) ; store C as bf16 / f32 / f16 (out_dtype = DCLIP_OUT_*; f16 needs act = DCLIP_ACT_NONE: the frozen teacher's in-place
 *   residual stream, reference _common.py:124-125 under `precision: 16`, config/final_config/l_clip.yaml:64).
 *   row_group > 0 adds `rowadd` (f32 [row_group, N]) row (r % row_group) to GEMM row r: the positional-embedding
 *   add of the token embedders (reference _common.py:196-202, text_encoder.py:65-66, weight_share_model.py:344-349,
 *   :487-489); for images the caller folds class token and conv bias into the table (see dclip_token_table).
 *   colsum_acc (f32 [N], may be NULL) += column sums of the stored values (bias gradient of the producing linear).
 */
int dclip_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                  int64_t M, int64_t N, int64_t K, float alpha, const float* bias, int act,
                  const void* aux_in, void* aux_out, const void* residual, int64_t ldr, int out_dtype,
                  int64_t row_group, const float* rowadd, float* colsum_acc, void* stream);

/*
 * dW[P,Q] (f32, ld = ldo) += sum_m A[m,P] * B[m,Q]         (weight gradient of nn.Linear: dY^T · X)
 *   autograd of the reference linears listed above (student only: weight_share_model.py:90,132,177,364).
 *   A bf16 [M,P] (lda), B bf16 [M,Q] (ldb).  P % 16 == 0, Q % 16 == 0.  Accumulates with f32 atomics
 *   (weight-shared layers add R uses per step, weight_share_model.py:199-218); `splits` >= 1 partitions M.
 */
/* workspace (nullable; dclip_gemm_tn_workspace_bytes() bytes, 16-byte aligned): with it the 256 x 256 wgrad pipeline writes each
 * split's partial tile with plain stores and a second launch adds the splits to dW in a fixed order (no f32 atomics: faster,
 * and run-to-run identical); without it, or for outputs on the small-tile kernels, f32 atomics as before. */
size_t dclip_gemm_tn_workspace_bytes(void);
/* diagnostic (process-global counter): how many 256 x 256 wgrad launches so far had to fall back from the partial-tile path (run-to-run
 * identical) to f32 atomics because no / too small a workspace was passed */
int64_t dclip_gemm_tn_atomic_fallbacks(void);
int dclip_gemm_tn_acc(const void* A, int64_t lda, const void* B, int64_t ldb, float* dW, int64_t ldo,
                      int64_t M, int64_t P, int64_t Q, int splits, void* workspace, size_t ws_bytes, void* stream);

/* db[N] (f32) += column sums of X[M,N] (bf16, ld) — bias gradient of nn.Linear. */
int dclip_colsum_acc(const void* X, int64_t ld, float* db, int64_t M, int64_t N, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * LayerNorm (fp32 statistics, eps inside the sqrt).
 *   reference: model/component/_common.py:14-20 (call sites :123,:125,:208,:210 ; text_encoder.py:69) and nn.LayerNorm in
 *   weight_share_model.py:181,183,363,503.
 * fwd: y[r] = LN(x[row_index ? row_index[r] : r]) * gamma + beta ; y is bf16 (out_f32=0) or f32 ; mean/rstd (f32 [M],
 *      nullable) are saved for backward.  D % 4 == 0, D <= 1024.
 * bwd: dx_acc[src(r)] += LN'(dy[r]) (f32, in place: the residual-stream gradient) ; optional bf16 copy of the updated
 *      rows in dx_bf16 ; dgamma / dbeta (f32 [D], nullable) += batch sums ; colsum_acc (f32 [D], nullable) += column sums
 *      of the updated dx_acc rows (the bias gradient of the linear that fed this residual stream).
 */
int dclip_layernorm_fwd(const float* x, int64_t ldx, const int32_t* row_index, const float* gamma, const float* beta,
                        void* y, int64_t ldy, int out_f32, float* mean, float* rstd, int64_t M, int64_t D, float eps,
                        void* stream);
/* the same on fp16 rows (x f16 [.., ldx]; y bf16 / f32 / f16 by out_dtype = DCLIP_OUT_*): the frozen teacher's LayerNorms, whose input is
 * the fp16 residual stream and which compute in f32 and return the input's type (reference _common.py:14-20) */
int dclip_layernorm_fwd_f16(const void* x, int64_t ldx, const int32_t* row_index, const float* gamma, const float* beta,
                            void* y, int64_t ldy, int out_dtype, float* mean, float* rstd, int64_t M, int64_t D, float eps,
                            void* stream);
int dclip_layernorm_bwd(const void* dy, int64_t lddy, int dy_f32, const float* x, int64_t ldx, const int32_t* row_index,
                        const float* gamma, const float* mean, const float* rstd, float* dx_acc, int64_t lddx,
                        void* dx_bf16, int64_t lddb, float* dgamma, float* dbeta, float* colsum_acc, int64_t M, int64_t D,
                        void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Attention building blocks.  q/k/v/ctx are token-major bf16 (row = b*N + n, column = head*hd + d, row stride ld*);
 * score-like tensors are [B,H,N,Np], Np = round_up(N, 8), pad columns zero.  hd in {32, 64}, N <= 128.
 *   reference teacher: _common.py:73-89 ; student: weight_share_model.py:101-125 (scale, QK^T, conv_l, softmax,
 *   conv_w, PV) ; causal mask: text_encoder.py:54-60.
 * nt : C[b,h,i,j] = alpha * sum_d A[(b,i),h*hd+d] * Bm[(b,j),h*hd+d]          (S = QK^T ; dR = dO V^T)
 * nn : C[(b,i),h*hd+d] = alpha * sum_j A[b,h,i,j] * Bm[(b,j),h*hd+d]          (O = R V ; dQ = dS K)
 * tn : C[(b,j),h*hd+d] = alpha * sum_i A[b,h,i,j] * Bm[(b,i),h*hd+d]          (dV = R^T dO ; dK = dS^T Q)
 * softmax_fwd : A_g = sum_h Wl[g,h] S_h ; P = softmax_j(A) (causal: j <= i) ; R_g = sum_h Ww[g,h] P_h.
 *               Wl/Ww NULL = plain multi-head softmax (CLIP towers; any head count — with mixing H must be 2, 4, 8, 12 or 24).
 *               P (nullable) is saved for backward.
 * softmax_bwd : dS from dR (+ dWl, dWw += H x H weight gradients, f32).
 */
int dclip_attn_nt(const void* A, int64_t lda, const void* Bm, int64_t ldb, void* C, int out_f32, int64_t B, int64_t H,
                  int64_t N, int64_t Np, int64_t hd, float alpha, void* stream);
/* a_blocked != 0: A is in the quad-blocked layout the register-resident score stage writes (dclip_attn_mix_fwd / _bwd):
 * [B, H, Np / 4, N, 4], element (i, j) of a (b, h) matrix at ((j >> 2) * N + i) * 4 + (j & 3); 0: row-major [B, H, N, Np]. */
int dclip_attn_nn(const void* A, const void* Bm, int64_t ldb, void* C, int64_t ldc, int64_t B, int64_t H, int64_t N,
                  int64_t Np, int64_t hd, float alpha, int a_blocked, void* stream);
int dclip_attn_tn(const void* A, const void* Bm, int64_t ldb, void* C, int64_t ldc, int64_t B, int64_t H, int64_t N,
                  int64_t Np, int64_t hd, float alpha, int a_blocked, void* stream);
/* fused_fwd : ctx = softmax(scale * q k^T (+ causal mask)) v for plain multi-head attention (teacher, _common.py:73-89);
 *              qkv is the fused [B*N, 3*H*hd] projection ; scores / probabilities never reach HBM. */
int dclip_attn_fused_fwd(const void* qkv, int64_t ldq, void* ctx, int64_t ldc, int64_t B, int64_t H, int64_t N, int64_t hd,
                         float scale, int causal, void* stream);
int dclip_attn_softmax_fwd(const float* S, const float* Wl, const float* Ww, void* P, void* R, int64_t B, int64_t H,
                           int64_t N, int64_t Np, int causal, void* stream);
int dclip_attn_softmax_bwd(const void* dR, const void* P, const void* S, int scores_bf16, const float* Wl, const float* Ww,
                           void* dS, float* dWl, float* dWw, int64_t B, int64_t H, int64_t N, int64_t Np, void* stream);
/*
 * Head-mixed student attention, score stage, with the score tensors kept in registers and BOTH head mixes on the matrix pipe
 * (attention_mix.hip + attn_mix_wave.h; reference weight_share_model.py:101-125).  One wave per (sample, 16 query rows):
 * S = scale q k^T comes out of block-diagonal MFMAs with lane group g holding head 4s + g, so that the packed accumulator
 * registers are the B operand of A = conv_l(S) with the mix matrix as the A operand; P = softmax(A) is a per-register exp2
 * against the log-sum-exp that enters as the initial accumulator; R = conv_w(P) contracts over the accumulator's row index
 * (no lane movement).  Only R (bf16, quad-blocked [B,H,Np/4,N,4]: element (i, j) at ((j >> 2) * N + i) * 4 + (j & 3), pad
 * columns zero -- a lane owns a query, so this is the layout in which a wave's 8-byte stores of 4 keys are contiguous over
 * the 16 queries of a tile; dclip_attn_nn / _tn read it with a_blocked = 1) and the softmax statistics (f32 [B,H,N]:
 * log-sum-exp of every row of A) are stored.  The backward recomputes S, A, P from the packed qkv rows, forms dR = dO v^T on the fly and
 * writes dS (bf16, same quad-blocked layout, gradient of the scaled pre-mix scores); dWl / dWw += [H,H] leave as one partial tile per workgroup in
 * `workspace` (dclip_attn_mix_bwd_workspace_bytes(B, H, N) bytes, 16-byte aligned; it also carries the [B,H,N] softmax-backward
 * row sums between the two launches of the backward) summed by a further launch: no atomics, run-to-run identical.  The products over keys / queries stay with dclip_attn_nn / dclip_attn_tn:
 *   forward   dclip_attn_mix_fwd -> dclip_attn_nn(R, v) ;   backward   dclip_attn_tn(R, dO) -> dV, dclip_attn_mix_bwd -> dS,
 *             dclip_attn_nn(dS, k) -> dQ, dclip_attn_tn(dS, q) -> dK.
 * Replaces dclip_attn_nt + dclip_attn_softmax_fwd and dclip_attn_nt + dclip_attn_softmax_bwd (S f32, P, dR never stored).
 * Mix operands are f16 in the forward (the precision of the reference's fp16 autocast) and bf16 on the gradient side.
 * dclip_attn_mix_supported: H in {2, 4, 8, 12} with hd in {32, 64}, or H = 24 with hd = 32; N <= 128 (other shapes use the
 * unfused kernels).
 */
int dclip_attn_mix_supported(int64_t H, int64_t N, int64_t hd);
size_t dclip_attn_mix_bwd_workspace_bytes(int64_t B, int64_t H, int64_t N);
void dclip_attn_mix_debug_stamps(void* fwd, void* bwd_a, void* bwd_b);   /* diagnostics: per-tile cycle counts of later launches, NULL = off */
int dclip_attn_mix_fwd(const void* qkv, int64_t ld, const float* Wl, const float* Ww, void* R, float* stats, int64_t B, int64_t H,
                       int64_t N, int64_t Np, int64_t hd, float scale, void* stream);
int dclip_attn_mix_bwd(const void* qkv, int64_t ld, const void* dO, int64_t ldo, const float* Wl, const float* Ww,
                       const float* stats, void* dS, float* dWl, float* dWw, void* workspace, size_t ws_bytes, int64_t B, int64_t H,
                       int64_t N, int64_t Np, int64_t hd, float scale, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Embedding-side helpers (HBM-bound).
 * cast_bf16            : f32 -> bf16 copy (per-step weight down-cast).
 * cast_transpose_bf16  : W f32 [R,C] -> Wb bf16 [R,C] (nullable) and Wt bf16 [C,R] (nullable; the dgrad operand).
 * im2row               : image f32 [B,C,res,res] -> bf16 rows [B*(G*G+cls_rows), C*p*p], G = res / p; the class-token row
 *                        is zero.  Conv2d(k=p,s=p) of _common.py:176,196 / timm PatchEmbed (weight_share_model.py:250).
 * token_table          : out[0] = pos[0] + cls ; out[n>=1] = pos[n] + bias  (cls NULL: out[n] = pos[n] + bias)
 *                        (_common.py:199-202 ; weight_share_model.py:346-349, :489) ; token_table_bwd is its adjoint given
 *                        tok_sum[n] = sum_b G[b,n,:] from batch_sum_acc.
 * embed_gather         : out[r] = table[ids[(r / N) * id_stride + r % N]] + pos[r % N] (text_encoder.py:65-66 ;
 *                        weight_share_model.py:487-489); id_stride = tokens per caption in `ids`, N <= id_stride = tokens used.
 * embed_scatter_add    : dtable[ids[r]] += dx[r] (f32 atomics; rows with the hot ids 0 / vocab-2 / vocab-1 = padding / SOT / EOT
 *                        of the clip.tokenize layout are reduced per block first instead of contending on three table rows).
 * pick_index           : idx[b] = b*N + argmax_n ids[b, 0..id_stride) (text_encoder.py:86, weight_share_model.py:506); ids NULL: b*N.
 * gather_rows          : out[r] = src[idx[r]] (f32).
 * adamw                : torch.optim.AdamW step on flat f32 buffers (distil_model.py:160-162, dual_distill_model.py:194-196).
 */
int dclip_cast_bf16(const float* src, void* dst, int64_t n, void* stream);
int dclip_cast_f16_f32(const void* src, float* dst, int64_t n, void* stream);   /* f16 -> f32 (teacher hidden-state export) */
/* dst += src (f32) ; optional bf16 copy of the updated dst ; optional column sums of src (row length D; then n % D == 0) */
int dclip_axpy_f32(float* dst, const float* src, void* dst_bf16, int64_t n, float* colsum_acc, int64_t D, void* stream);
int dclip_cast_transpose_bf16(const float* W, void* Wb, void* Wt, int64_t R, int64_t C, void* stream);
/* n jobs of the above in one launch (per-step refresh of a student tower's bf16 weight cache): host arrays of device pointers
 * and shapes; Wb[i] / Wt[i] nullable per job. */
int dclip_cast_transpose_bf16_multi(const float* const* W, void* const* Wb, void* const* Wt, const int64_t* R, const int64_t* C,
                                    int64_t n, void* stream);
int dclip_im2row(const float* img, void* rows, int64_t B, int64_t C, int64_t res, int64_t patch, int cls_rows, void* stream);
int dclip_token_table(const float* pos, const float* cls, const float* bias, float* out, int64_t ntok, int64_t D, void* stream);
int dclip_token_table_bwd(const float* tok_sum, float* dpos, float* dcls, float* dbias, int64_t ntok, int64_t D, int has_cls,
                          void* stream);
int dclip_batch_sum_acc(const float* G, float* out, int64_t B, int64_t N, int64_t D, void* stream);
int dclip_embed_gather(const int64_t* ids, int64_t id_stride, const float* table, const float* pos, void* out, int out_dtype,
                       int64_t rows, int64_t N, int64_t D, void* stream);
int dclip_embed_scatter_add(const int64_t* ids, const void* dx, int dx_f32, float* dtable, int64_t rows, int64_t D,
                            int64_t vocab, void* stream);
int dclip_pick_index(const int64_t* ids, int64_t id_stride, int32_t* idx, int64_t B, int64_t N, void* stream);
int dclip_gather_rows(const float* src, int64_t ld, const int32_t* idx, float* out, int64_t rows, int64_t D, void* stream);
int dclip_adamw(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                float weight_decay, int64_t step, int zero_grad, void* stream);   /* zero_grad: g := 0 once consumed */
/* the same step on `count` <= DCLIP_ADAMW_MAX_RANGES ranges in ONE launch (HOST arrays of device pointers and lengths; every range a
 * multiple of 4 elements, 16-byte aligned): the sharded data-parallel optimizer updates one owned slice per gradient bucket */
#define DCLIP_ADAMW_MAX_RANGES 24
int dclip_adamw_multi(float* const* p, float* const* g, float* const* m, float* const* v, const int64_t* n, int32_t count, float lr,
                      float beta1, float beta2, float eps, float weight_decay, int64_t step, int zero_grad, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Fused distillation loss, forward + backward.
 *   reference: model/_loss.py:118-202 (cal_tow_tower_loss / cal_one_tower_loss), model/component/clip_model.py:37-44,
 *   model/loss_component/{out_l1,out_cos,out_kl,out_ce,clip_cos_diff,hard_label,soft_label,logits_mse}.py.
 * s_*, t_*: student / teacher last_representation, f32 [B,E] (E % 16 == 0, E <= 1024, B <= 4096).
 * cfg (HOST pointer, 10 floats): w_out_l1, w_out_cos, w_out_kl, w_out_ce, w_cos_diff, w_hard_label, w_soft_label,
 *   w_logits_mse (each = loss_scale * percent of _loss.py:148-152,195-200; 0 disables), temperature, two_tower (0/1).
 *   two_tower = 1: loss = 0.5 * (image_tower + text_tower) + cross-modal terms ; 0: tower 0 only (s_txt etc. ignored).
 * out_scalars (device, 16 f32): [0] total ; [1..4] image out_l1, out_cos, out_kl, out_ce ; [5..8] text ; [9..12] cos_diff,
 *   hard_label, soft_label, logits_mse — raw (un-scaled) term values.
 * d_s_img / d_s_txt (f32 [B,E]): d total / d student embeddings (overwritten).  The [B,B] logits never reach HBM.
 */
size_t dclip_distill_loss_workspace(int64_t B, int64_t E);
int dclip_distill_loss(const float* s_img, const float* t_img, const float* s_txt, const float* t_txt, int64_t B, int64_t E,
                       const float* cfg, float* out_scalars, float* d_s_img, float* d_s_txt, void* workspace,
                       size_t ws_bytes, void* stream);
/*
 * Training image transform after decode / resize / crop: RandAugment -> ToTensor -> Normalize.
 *   reference: data/component/ms_coco.py:15-26 (transform chain), rand_augment.py:10-87 (_apply_op), :128-166 (op table,
 *   per-image loop), utils.py:11-12 (CLIP mean / std).  Pixel results equal Pillow's (the reference's backend) bit for bit.
 * images: uint8 [B,H,W,3] device (RGB, what np.asarray(PIL image) holds); ops: device [B, num_ops] records, applied in order:
 *   op 0 identity ; 1 nearest affine, c[0..5] = Pillow's 16.16 fixed-point coefficients (a0 a1 a2 a3 a4 a5 of Geometry.c
 *   affine_fixed) ; 2 integer shift, source = (x + c[0], y + c[1]) ; 3 brightness, 4 contrast, 5 sharpness with enhancement
 *   factor f ; 6 posterize with byte mask c[0] ; 7 autocontrast ; 8 equalize.   (distillclip_amd/augment.py builds the
 *   records from RandAugment's (op name, magnitude) draws.)
 * mean3 / std3: HOST float[3].  out: f32 [B,3,H,W] = (byte / 255 - mean) / std.  aug_out (nullable): the augmented bytes
 * [B,H,W,3].  workspace: dclip_augment_workspace(B,H,W) bytes (two byte images per sample), unused when num_ops == 0.
 */
typedef struct dclip_aug_op {
    int32_t op;
    int32_t c[6];
    float f;
} dclip_aug_op;
size_t dclip_augment_workspace(int64_t B, int64_t H, int64_t W);
int dclip_augment_normalize(const uint8_t* images, int64_t B, int64_t H, int64_t W, const dclip_aug_op* ops, int num_ops,
                            const float* mean3, const float* std3, float* out, uint8_t* aug_out, void* workspace,
                            size_t ws_bytes, void* stream);
/*
 * Resize(S) + CenterCrop(S) of a batch of decoded RGB images of mixed sizes, Pillow's antialiased
 * bilinear two-pass resample evaluated on the crop window, bit-exact.
 *   reference: data/component/ms_coco.py:16-17,23-24 (transforms.Resize(224) + transforms.CenterCrop(224) on PIL images).
 * packed: device bytes, image i = uint8 [height, width, 3] at desc[i].src_offset.  desc: device [B].  tables: device int32;
 * image i's block at tables + table_offset holds  hb[S][2] (first source column, count) , hk[S][ksize_h] (22-bit fixed-point
 * weights) , vb[S][2] (first row relative to row0, count) , vk[S][ksize_v]  for the S crop columns / rows — the values
 * Pillow's precompute_coeffs + normalize_coeffs_8bpc produce (distillclip_amd/augment.py:resample_tables builds and caches
 * them per source size).  workspace: image i's horizontal-pass scratch uint8 [nrows, S, 3] at temp_offset.
 * out: uint8 [B,S,S,3].
 */
typedef struct dclip_resize_desc {
    int64_t src_offset;
    int64_t table_offset;
    int64_t temp_offset;
    int32_t height, width;
    int32_t row0, nrows;
    int32_t ksize_h, ksize_v;
} dclip_resize_desc;
int dclip_resize_center_crop(const uint8_t* packed, const dclip_resize_desc* desc, const int32_t* tables, int64_t B, int64_t S,
                             uint8_t* out, void* workspace, size_t ws_bytes, void* stream);
/*
 * Validation retrieval metrics of one image -> caption logits matrix, without materialising it.
 *   reference: dual_distill_model.py:271-275 (norm_and_logits), :204-212 (log_diag_score), :220-224 (log_acc with
 *   torchmetrics accuracy(top_k = k) against labels arange(n)), k_list :87 ; distil_model.py:171-191, :224-231.
 * img, txt: f32 [n, E] device, un-normalised (rows are divided by their L2 norm, no epsilon, as the reference does);
 * ks: HOST array of nk <= 8 cut-offs.  out (device, nk + 2 f32): acc@ks[0..nk) (fraction of rows whose matching caption is
 * among the ks[i] highest logits of the row; strict comparison, a tie with the diagonal counts for the diagonal),
 * out[nk] = mean_i softmax(logits_i)[i], out[nk + 1] = mean_i logits_ii.  rank_out (device int32 [n], nullable) receives
 * the number of captions that beat the matching one.  Other pairings (student image x teacher text, ...) are further calls.
 */
size_t dclip_retrieval_metrics_workspace(int64_t n, int64_t E);
int dclip_retrieval_metrics(const float* img, const float* txt, int64_t n, int64_t E, const int32_t* ks, int nk,
                            float* out, int32_t* rank_out, void* workspace, size_t ws_bytes, void* stream);
/*
 * Row block of the same loss for data-parallel global negatives (Collective 2): the four inputs are the GATHERED
 * [B, E] embeddings of all ranks, this call owns rows [row0, row0 + rows) — its own samples — against all B columns.
 * d_s_img / d_s_txt: [rows, E] gradients of the owned samples (d global loss / d embedding).  out_scalars: this block's share of
 * every scalar: summing the 16 values over the ranks (one small all-reduce) gives the loss of the concatenated batch.
 * hard_label / soft_label need the softmax statistics of EVERY row (both directions): call once with stats_out (f32 [6, rows]; only the
 * statistics pass runs, no gradients), all-gather the ranks' blocks into [6, B] (statistic-major, rows in rank order), and call again
 * with gathered_stats.  Both pointers null: only terms without such statistics (cos_diff, logits_mse, tower terms) may be enabled.
 * Same workspace query (with the gathered B).
 */
int dclip_distill_loss_rows(const float* s_img, const float* t_img, const float* s_txt, const float* t_txt, int64_t B, int64_t E,
                            int64_t row0, int64_t rows, const float* cfg, float* out_scalars, float* d_s_img, float* d_s_txt,
                            const float* gathered_stats, float* stats_out, void* workspace, size_t ws_bytes, void* stream);
/* feature MSE (hidden_rep_mse / embedding_mse: hidden_mse.py:9-17, embed_mse.py:9-10):
 *   loss_acc[0] += coef * mean((s - t)^2) ; ds_acc (nullable) += coef * 2 (s - t) / n */
int dclip_feature_mse(const float* s, const float* t, int64_t n, float coef, float* loss_acc, float* ds_acc, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Tower-level runtime: one call issues the whole launch sequence of an encoder tower on `stream`.
 *   teacher: reference model/component/_common.py:188-221 (VisionTransformer.forward) and
 *            model/component/text_encoder.py:62-92 (TextEncoder.encode_text) — kind 0: frozen, inference only, fp16 residual stream;
 *            kind 2: the same architecture and parameter order as a TRAINABLE tower — the plain ImageEncoder / TextEncoder in the
 *            student role (reference image_encoder.py:16-25,54-59, text_encoder.py:41-47,75-80; their embedding_projection /
 *            hidden_projection linears act on the exported hidden states and stay with the caller) — f32 residual stream, autograd of it;
 *   student: reference model/component/weight_share_model.py:336-372 / :482-512 (forward_features) and autograd of it.
 * The handle is an immutable host-side plan; params / grads / wcache / workspace are caller-owned device buffers.
 *
 * Canonical parameter order (arrays of f32 device pointers; names are the reference's state_dict keys):
 *  teacher image : visual.conv1.weight, visual.class_embedding, visual.positional_embedding, visual.ln_pre.{weight,bias},
 *                  per layer i { ln_1.weight, ln_1.bias, attn.in_proj_weight, attn.in_proj_bias, attn.out_proj.weight,
 *                  attn.out_proj.bias, ln_2.weight, ln_2.bias, mlp.c_fc.weight, mlp.c_fc.bias, mlp.c_proj.weight,
 *                  mlp.c_proj.bias }, visual.ln_post.{weight,bias}, visual.proj
 *  teacher text  : token_embedding.weight, positional_embedding, per layer i { same 12 }, ln_final.{weight,bias},
 *                  text_projection
 *  student image : patch_embed.proj.weight, patch_embed.proj.bias, cls_token, pos_embed,
 *                  per block i { attn.qkv.weight, attn.qkv.bias (NULL if absent), attn.proj.weight, attn.proj.bias,
 *                  mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias, per repeat r { norm1.instances.r.weight,
 *                  .bias, norm2.instances.r.weight, .bias, attn.conv_l.instances.r.weight, attn.conv_w.instances.r.weight
 *                  (NULL, NULL when head_mix = 0) } }, norm.weight, norm.bias, head.weight, head.bias
 *  student text  : patch_embed.weight, pos_embed   (embed_rank = 0)   or
 *                  patch_embed.0.weight, patch_embed.1.weight, patch_embed.1.bias, pos_embed   (embed_rank > 0),
 *                  then blocks / norm / head as above.
 * `grads` uses the same order; a NULL entry means "frozen, skip" (requires_grad = False).  Gradients ACCUMULATE (+=).
 */
typedef struct dclip_encoder_cfg {
    int32_t kind;        /* 0 teacher (CLIP residual blocks, QuickGELU; frozen), 1 student (weight-shared MiniViT blocks, erf GELU),
                          * 2 CLIP tower that trains (architecture and parameter order of kind 0, workspace / backward of kind 1) */
    int32_t modality;    /* 0 image, 1 text */
    int32_t tokens;      /* N: (resolution / patch)^2 + 1 for images, context_length for text */
    int32_t width;       /* D */
    int32_t heads;       /* H (D / H in {32, 64}) */
    int32_t layers;      /* distinct blocks: teacher = transformer layers, student = depth / repeated_times */
    int32_t repeats;     /* repeated_times (teacher: 1) */
    int32_t mlp_dim;     /* int(D * mlp_ratio) */
    int32_t out_dim;     /* E */
    int32_t patch, resolution, in_chans;  /* image only; resolution = height = width of the INPUT images; the patch conv floors
                                           * (grid = resolution / patch: 336 px at patch 32 reads the top-left 320 x 320) */
    int32_t vocab, embed_rank;            /* text only; embed_rank = embedding_compression_dim or 0 */
    int32_t head_mix;    /* use_transform: conv_l / conv_w cross-head mixing */
    int32_t causal;      /* CLIP text towers: 1 */
} dclip_encoder_cfg;

typedef struct dclip_encoder dclip_encoder;

dclip_encoder* dclip_encoder_create(const dclip_encoder_cfg* cfg);   /* NULL + last_error_string on a bad configuration */
void dclip_encoder_destroy(dclip_encoder* enc);
int64_t dclip_encoder_num_params(const dclip_encoder* enc);
size_t dclip_encoder_wcache_bytes(const dclip_encoder* enc);
size_t dclip_encoder_workspace_bytes(const dclip_encoder* enc, int64_t B, int training);
/* refresh the bf16 GEMM-weight cache from the f32 parameters (student: every step; teacher: once) */
int dclip_encoder_prepare(const dclip_encoder* enc, const void* const* params, void* wcache, void* stream);
/* input: image f32 [B,C,res,res] or token ids i64 [B,N].  last_representation: f32 [B,E] (class token / EOT row).
 * training = 1 keeps every activation backward needs inside `workspace` (kinds 1 and 2).
 * rep_out (nullable array of layers*repeats nullable f32 [B*N, D] pointers) / emb_out (nullable f32 [B*N, D]) receive the hidden
 * state after each block execution and the post-positional-embedding tokens (ControlOutput.need_rep / need_emb of the
 * reference, _loss.py:100-116); d_rep / d_emb are the matching gradients, added to the residual-stream gradient in backward.
 * tokens_eff (0 = all): causal text teacher only — run the tower on the first tokens_eff positions of every caption.  The
 * caller guarantees that every caption's EOT lies inside that prefix; positions after it cannot influence the EOT row
 * (causal mask), so last_representation is unchanged. */
int dclip_encoder_forward(const dclip_encoder* enc, const void* input, int64_t B, const void* const* params,
                          const void* wcache, void* workspace, size_t ws_bytes, int training, float* last_representation,
                          float* const* rep_out, float* emb_out, int64_t tokens_eff, void* stream);
/* The same forward on patch rows the caller has already cut: `patches` = bf16 [B*N, C*patch*patch] from dclip_im2row(..., cls_rows = 1).
 * Teacher and student see the same image batch (reference dual_distill_model.py:107-109, distil_model.py forward) and, when their
 * patch size and resolution agree, the same conv1 / PatchEmbed unfolding (_common.py:196-198, weight_share_model.py:344): one
 * conversion then serves both towers.  Image towers only; the matching backward is dclip_encoder_backward_patches. */
int dclip_encoder_forward_patches(const dclip_encoder* enc, const void* patches, int64_t B, const void* const* params,
                                  const void* wcache, void* workspace, size_t ws_bytes, int training, float* last_representation,
                                  float* const* rep_out, float* emb_out, void* stream);
/* last_layer_output (reference output.py:16-35; _common.py:210-215, text_encoder.py:69-72, weight_share_model.py:363-366,
 * :503-506): final norm + projection of EVERY token, f32 [B*N, E], computed on request from the residual stream the most
 * recent dclip_encoder_forward(enc, ..., B, training) left in `workspace` (tokens_eff must have been 0).  scratch: bf16
 * [B*N, D] caller-owned.  last_representation is the class-token / EOT row of this tensor. */
int dclip_encoder_last_layer_output(const dclip_encoder* enc, int64_t B, const void* const* params, const void* wcache,
                                    void* workspace, size_t ws_bytes, int training, void* scratch, float* out, void* stream);
/* on_bucket (nullable): host callback, invoked on the calling thread as soon as every launch that writes gradient bucket
 * `bucket` has been enqueued on `stream` (an event recorded on `stream` inside the callback marks the bucket complete): the
 * data-parallel exchange of that bucket may start while the rest of the backward runs (reference: Lightning DDP's bucketed
 * all-reduce from autograd hooks, config/final_config/l_clip.yaml:56 strategy ddp_find_unused_parameters_false).
 * Buckets complete in index order; see dclip_encoder_grad_bucket.
 * Seeds: the backward starts from a residual-stream gradient accumulator (and one bf16 operand slot) that must be zero.  The training
 * forward clears them at its end (beside the other towers' work) and the handle remembers the workspace it did that for; a backward that
 * does not find ITS workspace there — a second backward on one forward, a retry, another workspace used in between — clears them itself.
 * Either way one call = the gradients of the most recent training forward of that workspace for the given d_*; gradients ACCUMULATE (+=)
 * into `grads` as everywhere. */
typedef void (*dclip_bucket_cb)(void* user, int32_t bucket);
int dclip_encoder_backward(const dclip_encoder* enc, const void* input, int64_t B, const void* const* params,
                           void* const* grads, const void* wcache, void* workspace, size_t ws_bytes,
                           const float* d_last_representation, const float* const* d_rep, const float* d_emb,
                           dclip_bucket_cb on_bucket, void* cb_user, void* stream);
/* backward of a forward that ran on caller-made patch rows (they are the patch-embedding wgrad's operand) */
int dclip_encoder_backward_patches(const dclip_encoder* enc, const void* patches, int64_t B, const void* const* params,
                                   void* const* grads, const void* wcache, void* workspace, size_t ws_bytes,
                                   const float* d_last_representation, const float* const* d_rep, const float* d_emb,
                                   dclip_bucket_cb on_bucket, void* cb_user, void* stream);
/* gradient buckets in completion order: 0 = final norm + head, 1..L = blocks L-1..0, L+1 = embedding parameters; each is the
 * range [first_param, end_param) of the canonical parameter order above. */
int32_t dclip_encoder_num_grad_buckets(const dclip_encoder* enc);
int dclip_encoder_grad_bucket(const dclip_encoder* enc, int32_t bucket, int32_t* first_param, int32_t* end_param);

/* ---------------------------------------------------------------------------------------------------------------
 * Launch trace (profiling only; process-global): between begin and end every GEMM / LayerNorm-forward / loss call is
 * bracketed by HIP events on the stream it is launched on.  trace_end writes (kind, ms, algorithmic flops, algorithmic
 * bytes) per call and returns the number of calls seen.  kind: 0 gemm_nt, 1 gemm_tn_acc, 2 layernorm_fwd, 3 distill_loss,
 * 4 attention family (nt / nn / tn / fused / softmax fwd / softmax bwd), 5 layernorm_bwd. */
int dclip_trace_begin(int64_t max_records);
int64_t dclip_trace_end(int32_t* kind, float* ms, double* flops, double* bytes, int64_t cap);
/* problem sizes of the traced calls so far, 4 ints per record (GEMM: M, N, K, variant bits); call before dclip_trace_end */
int64_t dclip_trace_dims(int32_t* dims, int64_t cap);
/* profiling only (process-global): while `buf` is non-NULL every workgroup of the 256- / 320-row dclip_gemm_nt kernels writes 6
 * uint64 stamps to buf[6 * workgroup ..]: s_memtime at start / first operands landed / main loop done / epilogue done, then
 * s_memrealtime (100 MHz) at start / end (tools/diag/gemm_phases.py).  NULL switches it off. */
int dclip_trace_gemm_stamps(void* buf);
/* same for the head-mixing softmax backward: 8 uint64 per (wave, row iteration < 4): s_memtime at row start / operands in LDS /
 * row sums done / key tiles done / dW_l done (tools/diag/attn_phases.py) */
int dclip_trace_attn_stamps(void* buf);
/* measurement only (bench.py `clock_mhz_during_timed_steps`; process-global): while `buf` is non-NULL, workgroup 0 of every 256- / 320-row
 * dclip_gemm_nt launch writes 4 uint64 — s_memtime (shader cycles) and s_memrealtime (100 MHz) at its start and at its end — to
 * buf[4 * (launch % cap) ..]; shader clock held during that launch = d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS
 * give-back).  One thread of one workgroup per launch: the timed steps are not perturbed (a resident probe wave on a stream of its own
 * cost 1.7 % of the step and was dropped).  Returns the number of launches stamped since the previous call; NULL switches it off. */
int64_t dclip_trace_gemm_clock(void* buf, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* DCLIP_H */
