/*
 * dclip.h — C ABI of libdistillclip_hip.so (MI355X / gfx950).
 *
 * The reference (ForJadeForest/DistillCLIP) is 100 % Python on top of PyTorch ATen; it has no FFI of its own.
 * The boundary this library replaces is therefore the implicit ATen kernel sequence behind the reference's
 * nn.Module calls.  Each entry point cites the reference call site (file:line under the reference root) whose
 * arithmetic it implements.  All pointers are raw device pointers; `stream` is a hipStream_t passed as void*.
 * No allocation, no synchronisation and no global mutable state inside any entry point; every function returns
 * 0 on success, DCLIP_EINVAL (-1) for a bad argument, DCLIP_ELAUNCH (-2) for a HIP launch failure, and
 * dclip_last_error_string() (thread-local) explains the last failure.
 *
 * dtypes: "bf16" = bfloat16 storage (MFMA operands), "f32" = float.  Row-major everywhere.
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
const char* dclip_arch(void);               /* "gfx950" */
const char* dclip_last_error_string(void);  /* thread-local */

/* activation codes for dclip_gemm_nt */
#define DCLIP_ACT_NONE 0
#define DCLIP_ACT_QUICKGELU 1 /* reference model/component/_common.py:23-25 */
#define DCLIP_ACT_GELU 2      /* exact erf GELU: timm Mlp act, reference weight_share_model.py:177 */
#define DCLIP_ACT_DGELU 3     /* multiply by gelu'(aux_in): backward of DCLIP_ACT_GELU */

/*
 * C[M,N] = epilogue(alpha * A[M,K] · B[N,K]^T)        (nn.Linear / F.linear / x @ proj / conv-as-GEMM)
 *   reference: _common.py:59,90,104-108,213 ; text_encoder.py:72 ; weight_share_model.py:90,132,177,364
 *   A, B bf16 (lda, ldb in elements; K % 64 == 0; 16-byte aligned rows).
 *   epilogue, in order: + bias[N] (f32, may be NULL) ; if aux_out: store pre-activation as bf16 [M,N] (ld = ldc) ;
 *   activation `act` (DCLIP_ACT_DGELU multiplies by gelu'(aux_in[M,N] bf16, ld = ldc)) ;
 *   + residual[M,N] (f32, ld = ldr, may be NULL, may alias C when out_f32) ; store C as f32 (out_f32=1) or bf16.
 *   row_group > 0 enables the patch-embedding row map (reference _common.py:196-202, weight_share_model.py:344-349):
 *   GEMM row r is stored at row r + r / row_group + 1 and `rowadd` (f32 [row_group + 1, N], the positional
 *   embedding) row (r % row_group) + 1 is added.
 */
int dclip_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                  int64_t M, int64_t N, int64_t K, float alpha, const float* bias, int act,
                  const void* aux_in, void* aux_out, const float* residual, int64_t ldr, int out_f32,
                  int64_t row_group, const float* rowadd, void* stream);

/*
 * dW[P,Q] (f32, ld = ldo) += sum_m A[m,P] * B[m,Q]         (weight gradient of nn.Linear: dY^T · X)
 *   autograd of the reference linears listed above (student only: weight_share_model.py:90,132,177,364).
 *   A bf16 [M,P] (lda), B bf16 [M,Q] (ldb).  P % 16 == 0, Q % 16 == 0.  Accumulates with f32 atomics
 *   (weight-shared layers add R uses per step, weight_share_model.py:199-218); `splits` >= 1 partitions M.
 */
int dclip_gemm_tn_acc(const void* A, int64_t lda, const void* B, int64_t ldb, float* dW, int64_t ldo,
                      int64_t M, int64_t P, int64_t Q, int splits, void* stream);

/* db[N] (f32) += column sums of X[M,N] (bf16, ld) — bias gradient of nn.Linear. */
int dclip_colsum_acc(const void* X, int64_t ld, float* db, int64_t M, int64_t N, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCLIP_H */
