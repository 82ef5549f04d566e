"""Thin torch-tensor wrappers over the C ABI (device pointers + current stream).  Plumbing only: every
function launches hand-written HIP kernels from libdistillclip_hip.so and raises if the tensors are not on a GPU."""
import torch

from ._lib import lib

ACT = {'none': 0, None: 0, 'quickgelu': 1, 'gelu': 2, 'dgelu': 3}


def _p(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('distillclip_amd ops need CUDA(HIP) tensors; there is no CPU fallback')


def gemm_nt(a, b, *, bias=None, act=None, aux_in=None, aux_out=None, residual=None, out=None, out_dtype=torch.bfloat16,
            alpha=1.0, row_group=0, rowadd=None, out_rows=None):
    """out[M,N] = epilogue(alpha * a[M,K] @ b[N,K]^T); see include/dclip.h:dclip_gemm_nt."""
    _chk(a, b, bias, aux_in, aux_out, residual, out, rowadd)
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.dim() == 2 and b.dim() == 2
    assert a.stride(1) == 1 and b.stride(1) == 1
    M, K = a.shape
    N = b.shape[0]
    if out is None:
        rows = out_rows if out_rows is not None else M
        out = torch.empty((rows, N), dtype=out_dtype, device=a.device)
    assert out.stride(1) == 1
    ldr = residual.stride(0) if residual is not None else 0
    lib().dclip_gemm_nt(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N, K, float(alpha),
                        _p(bias), ACT[act], _p(aux_in), _p(aux_out), _p(residual), ldr,
                        1 if out.dtype == torch.float32 else 0, row_group, _p(rowadd), _stream())
    return out


def gemm_tn_acc(a, b, dw, splits=4):
    """dw[P,Q] (f32) += a[M,P]^T @ b[M,Q]."""
    _chk(a, b, dw)
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and dw.dtype == torch.float32
    M, P = a.shape
    Q = b.shape[1]
    assert b.shape[0] == M and tuple(dw.shape) == (P, Q) and dw.stride(1) == 1
    lib().dclip_gemm_tn_acc(_p(a), a.stride(0), _p(b), b.stride(0), _p(dw), dw.stride(0), M, P, Q, splits, _stream())
    return dw


def colsum_acc(x, db):
    _chk(x, db)
    assert x.dtype == torch.bfloat16 and db.dtype == torch.float32
    lib().dclip_colsum_acc(_p(x), x.stride(0), _p(db), x.shape[0], x.shape[1], _stream())
    return db
