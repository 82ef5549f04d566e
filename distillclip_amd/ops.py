"""Thin torch-tensor wrappers over the C ABI (device pointers + current stream).  Plumbing only: every
function launches hand-written HIP kernels from libdistillclip_hip.so and raises if the tensors are not on a GPU."""
import torch

from ._lib import lib

ACT = {'none': 0, None: 0, 'quickgelu': 1, 'gelu': 2, 'dgelu': 3, 'mulaux': 4, 'gelu_save': 5, 'quickgelu_save': 6}
OUT_DTYPE = {torch.bfloat16: 0, torch.float32: 1, torch.float16: 2}      # DCLIP_OUT_*
# 8-bit fixed-point code of the saved gelu' (include/dclip.h: DCLIP_ACT_GELU_SAVE / DCLIP_ACT_MULAUX): value = DG_LO + q * DG_STEP
DG_LO, DG_STEP = -0.13, 1.26 / 255.0


def _p(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('distillclip_amd ops need CUDA(HIP) tensors; there is no CPU fallback')


def gemm_nt(a, b, *, bias=None, act=None, aux_in=None, aux_out=None, residual=None, out=None, out_dtype=torch.bfloat16,
            alpha=1.0, row_group=0, rowadd=None, out_rows=None, colsum=None):
    """out[M,N] = epilogue(alpha * a[M,K] @ b[N,K]^T); see include/dclip.h:dclip_gemm_nt.
    act='gelu_save' / 'quickgelu_save' write, act='mulaux' reads the 8-bit derivative (uint8 [M,N]); out_dtype float16 = the teacher's fp16 residual stream
    (residual then float16 too, act none)."""
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
    if act in ('gelu_save', 'quickgelu_save') and aux_out is not None:
        assert aux_out.dtype == torch.uint8 and aux_out.stride(0) == out.stride(0)
    if act == 'mulaux':
        assert aux_in.dtype == torch.uint8 and aux_in.stride(0) == out.stride(0)
    if residual is not None:
        assert residual.dtype == (torch.float16 if out.dtype == torch.float16 else torch.float32)
    lib().dclip_gemm_nt(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N, K, float(alpha),
                        _p(bias), ACT[act], _p(aux_in), _p(aux_out), _p(residual), ldr,
                        OUT_DTYPE[out.dtype], row_group, _p(rowadd), _p(colsum), _stream())
    return out


_TN_WS = {}


def _tn_workspace(device):
    # one workspace per (device, stream): two wgrad calls enqueued on different streams must not share partial tiles
    key = (device, _stream())
    ws = _TN_WS.get(key)
    if ws is None:
        ws = _TN_WS[key] = torch.empty(lib().dclip_gemm_tn_workspace_bytes(), dtype=torch.uint8, device=device)
    return ws


def gemm_tn_acc(a, b, dw, splits=4, workspace=True):
    """dw[P,Q] (f32) += a[M,P]^T @ b[M,Q].  workspace=True: large outputs leave as per-split partial tiles + a fixed-order sum
    (no f32 atomics, run-to-run identical); False: f32 atomics."""
    _chk(a, b, dw)
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and dw.dtype == torch.float32
    M, P = a.shape
    Q = b.shape[1]
    assert b.shape[0] == M and tuple(dw.shape) == (P, Q) and dw.stride(1) == 1
    ws = _tn_workspace(a.device) if workspace else None
    lib().dclip_gemm_tn_acc(_p(a), a.stride(0), _p(b), b.stride(0), _p(dw), dw.stride(0), M, P, Q, splits, _p(ws),
                            0 if ws is None else ws.numel(), _stream())
    return dw


def colsum_acc(x, db):
    _chk(x, db)
    assert x.dtype == torch.bfloat16 and db.dtype == torch.float32
    lib().dclip_colsum_acc(_p(x), x.stride(0), _p(db), x.shape[0], x.shape[1], _stream())
    return db


def layernorm_fwd(x, gamma, beta, *, row_index=None, out_dtype=torch.bfloat16, eps=1e-5, save_stats=True):
    """x f32, or f16 (the frozen teacher's residual stream: dclip_layernorm_fwd_f16, which can also return f16)"""
    _chk(x, gamma, beta, row_index)
    assert x.dtype in (torch.float32, torch.float16) and x.dim() == 2 and x.stride(1) == 1
    M = row_index.numel() if row_index is not None else x.shape[0]
    D = x.shape[1]
    y = torch.empty((M, D), dtype=out_dtype, device=x.device)
    mean = torch.empty(M, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(M, dtype=torch.float32, device=x.device) if save_stats else None
    fn = lib().dclip_layernorm_fwd_f16 if x.dtype == torch.float16 else lib().dclip_layernorm_fwd
    fn(_p(x), x.stride(0), _p(row_index), _p(gamma), _p(beta), _p(y), D, OUT_DTYPE[out_dtype], _p(mean), _p(rstd), M, D, eps, _stream())
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dx_acc, *, row_index=None, dx_bf16=None, dgamma=None, dbeta=None, colsum=None):
    _chk(dy, x, gamma, mean, rstd, dx_acc, dx_bf16, dgamma, dbeta)
    M, D = dy.shape
    lib().dclip_layernorm_bwd(_p(dy), dy.stride(0), 1 if dy.dtype == torch.float32 else 0, _p(x), x.stride(0),
                              _p(row_index), _p(gamma), _p(mean), _p(rstd), _p(dx_acc), dx_acc.stride(0), _p(dx_bf16),
                              dx_bf16.stride(0) if dx_bf16 is not None else 0, _p(dgamma), _p(dbeta), _p(colsum), M, D, _stream())


def attn_nt(a, lda, bm, ldb, B, H, N, hd, alpha=1.0, out_dtype=torch.float32):
    Np = (N + 7) // 8 * 8
    c = torch.empty((B, H, N, Np), dtype=out_dtype, device=a.device)
    lib().dclip_attn_nt(_p(a), lda, _p(bm), ldb, _p(c), 1 if out_dtype == torch.float32 else 0, B, H, N, Np, hd, alpha,
                        _stream())
    return c


def _score_dims(a):
    """row-major [B,H,N,Np] or the quad-blocked [B,H,Np/4,N,4] of the register-resident score stage -> (B, H, N, Np, blocked)"""
    if a.dim() == 5:
        B, H, nq, N, four = a.shape
        assert four == 4
        return B, H, N, nq * 4, 1
    B, H, N, Np = a.shape
    return B, H, N, Np, 0


def attn_nn(a, bm, ldb, c, ldc, hd, alpha=1.0):
    B, H, N, Np, blocked = _score_dims(a)
    lib().dclip_attn_nn(_p(a), _p(bm), ldb, _p(c), ldc, B, H, N, Np, hd, alpha, blocked, _stream())


def attn_tn(a, bm, ldb, c, ldc, hd, alpha=1.0):
    B, H, N, Np, blocked = _score_dims(a)
    lib().dclip_attn_tn(_p(a), _p(bm), ldb, _p(c), ldc, B, H, N, Np, hd, alpha, blocked, _stream())


def unblock_scores(a):
    """quad-blocked [B,H,Np/4,N,4] -> row-major [B,H,N,Np] (tests / diagnostics)"""
    B, H, nq, N, _ = a.shape
    return a.permute(0, 1, 3, 2, 4).reshape(B, H, N, nq * 4)


def attn_softmax_fwd(s, wl=None, ww=None, causal=False, save_p=False):
    B, H, N, Np = s.shape
    r = torch.empty((B, H, N, Np), dtype=torch.bfloat16, device=s.device)
    p = torch.empty_like(r) if save_p else None
    lib().dclip_attn_softmax_fwd(_p(s), _p(wl), _p(ww), _p(p), _p(r), B, H, N, Np, 1 if causal else 0, _stream())
    return p, r


def attn_softmax_bwd(dr, p, s, wl=None, ww=None, dwl=None, dww=None):
    B, H, N, Np = dr.shape
    ds = torch.empty_like(dr)
    lib().dclip_attn_softmax_bwd(_p(dr), _p(p), _p(s), 1 if (s is not None and s.dtype == torch.bfloat16) else 0, _p(wl), _p(ww), _p(ds),
                                 _p(dwl), _p(dww), B, H, N, Np, _stream())
    return ds


LOSS_SLOTS = {'out_l1': 0, 'out_cos': 1, 'out_kl': 2, 'out_ce': 3, 'cos_diff': 4, 'hard_label': 5, 'soft_label': 6,
              'logits_mse': 7}


def distill_loss(s_img, t_img, s_txt=None, t_txt=None, *, weights, temperature=None, row0=None, rows=None, gathered_stats=None,
                 stats_only=False):
    """Fused loss fwd+bwd.  weights: {term: scale*percent}.  -> (scalars[16] device tensor, d_s_img, d_s_txt).
    row0 / rows: evaluate only the row block [row0, row0 + rows) of the (gathered) batch against all columns — gradients
    [rows, E] of the owned samples, scalars = this block's share (sum over the blocks = the whole-batch values).
    hard_label / soft_label in row-block mode: first `stats_only=True` -> [6, rows] statistics of the owned rows; gather all
    blocks into [6, B] and pass them as `gathered_stats` to the second call."""
    import ctypes
    _chk(s_img, t_img, s_txt, t_txt)
    B, E = s_img.shape
    two = s_txt is not None
    cfg = [0.0] * 10
    for k, w in weights.items():
        cfg[LOSS_SLOTS[k]] = float(w)
    cfg[8] = float(temperature or 0.0)
    cfg[9] = 1.0 if two else 0.0
    cfg_arr = (ctypes.c_float * 10)(*cfg)
    ws_bytes = lib().dclip_distill_loss_workspace(B, E)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=s_img.device)
    out = torch.empty(16, dtype=torch.float32, device=s_img.device)
    nrow = B if rows is None else int(rows)
    d_i = torch.empty((nrow, E), dtype=torch.float32, device=s_img.device)
    d_t = torch.empty((nrow, E), dtype=torch.float32, device=s_img.device) if two else None
    if rows is None:
        lib().dclip_distill_loss(_p(s_img), _p(t_img), _p(s_txt), _p(t_txt), B, E, ctypes.cast(cfg_arr, ctypes.c_void_p),
                                 _p(out), _p(d_i), _p(d_t), _p(ws), ws_bytes, _stream())
    else:
        stats = torch.empty((6, nrow), dtype=torch.float32, device=s_img.device) if stats_only else None
        if gathered_stats is not None:
            assert gathered_stats.shape == (6, B) and gathered_stats.dtype == torch.float32 and gathered_stats.is_contiguous()
        lib().dclip_distill_loss_rows(_p(s_img), _p(t_img), _p(s_txt), _p(t_txt), B, E, int(row0), nrow,
                                      ctypes.cast(cfg_arr, ctypes.c_void_p), _p(out), _p(d_i), _p(d_t), _p(gathered_stats),
                                      _p(stats), _p(ws), ws_bytes, _stream())
        if stats_only:
            return stats
    return out, d_i, d_t


def attn_fused_fwd(qkv, B, N, H, hd, causal=False):
    """ctx = softmax(q k^T / sqrt(hd) (+ causal mask)) v for plain multi-head attention; qkv: [B*N, 3*H*hd] bf16."""
    _chk(qkv)
    D = H * hd
    ctx = torch.empty((B * N, D), dtype=torch.bfloat16, device=qkv.device)
    lib().dclip_attn_fused_fwd(_p(qkv), qkv.stride(0), _p(ctx), D, B, H, N, hd, hd ** -0.5, 1 if causal else 0, _stream())
    return ctx


def embed_scatter_add(ids, dx, dtable):
    _chk(ids, dx, dtable)
    rows, D = dx.shape
    lib().dclip_embed_scatter_add(_p(ids), _p(dx), 1 if dx.dtype == torch.float32 else 0, _p(dtable), rows, D, dtable.shape[0],
                                  _stream())
    return dtable


def feature_mse(s, t, coef=1.0):
    """-> (mean((s - t)^2) * coef as a 0-d tensor, d/ds of it)"""
    _chk(s, t)
    assert s.dtype == torch.float32 and t.dtype == torch.float32 and s.numel() == t.numel() and s.numel() % 4 == 0
    val = torch.zeros(1, dtype=torch.float32, device=s.device)
    ds = torch.zeros_like(s)
    lib().dclip_feature_mse(_p(s), _p(t), s.numel(), float(coef), _p(val), _p(ds), _stream())
    return val[0], ds


def attn_mix_fwd(qkv, B, N, H, hd, wl, ww, scale):
    """register-resident head-mixed attention scores (include/dclip.h: dclip_attn_mix_fwd)
    -> (R bf16, quad-blocked [B,H,Np/4,N,4] (unblock_scores() gives [B,H,N,Np]), lse f32 [B,H,N])"""
    _chk(qkv, wl, ww)
    Np = (N + 7) // 8 * 8
    r = torch.full((B, H, Np // 4, N, 4), float('nan'), dtype=torch.bfloat16, device=qkv.device)    # the kernel writes every element
    stats = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    lib().dclip_attn_mix_fwd(_p(qkv), qkv.stride(0), _p(wl), _p(ww), _p(r), _p(stats), B, H, N, Np, hd, scale, _stream())
    return r, stats


def attn_mix_bwd(qkv, d_ctx, B, N, H, hd, wl, ww, stats, scale, dwl, dww):
    """-> dS bf16, quad-blocked [B,H,Np/4,N,4] (gradient of the scaled pre-mix scores); dwl / dww += (include/dclip.h: dclip_attn_mix_bwd)"""
    _chk(qkv, d_ctx, wl, ww, stats, dwl, dww)
    Np = (N + 7) // 8 * 8
    ds = torch.full((B, H, Np // 4, N, 4), float('nan'), dtype=torch.bfloat16, device=qkv.device)  # the kernel writes every element
    ws = torch.empty(lib().dclip_attn_mix_bwd_workspace_bytes(B, H, N), dtype=torch.uint8, device=qkv.device)
    lib().dclip_attn_mix_bwd(_p(qkv), qkv.stride(0), _p(d_ctx), d_ctx.stride(0), _p(wl), _p(ww), _p(stats), _p(ds), _p(dwl), _p(dww),
                             _p(ws), ws.numel(), B, H, N, Np, hd, scale, _stream())
    return ds


def cast_transpose_multi(ws, want_b=True, want_t=True):
    """[W f32 [R, C]] -> ([bf16 [R, C]] or None, [bf16 [C, R]] or None) in ONE launch (include/dclip.h: dclip_cast_transpose_bf16_multi)"""
    import ctypes
    _chk(*ws)
    n = len(ws)
    wb = [torch.empty_like(w, dtype=torch.bfloat16) for w in ws] if want_b else [None] * n
    wt = [torch.empty((w.shape[1], w.shape[0]), dtype=torch.bfloat16, device=w.device) for w in ws] if want_t else [None] * n
    arr = lambda ts: (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in ts])
    R = (ctypes.c_int64 * n)(*[w.shape[0] for w in ws])
    C = (ctypes.c_int64 * n)(*[w.shape[1] for w in ws])
    lib().dclip_cast_transpose_bf16_multi(arr(ws), arr(wb), arr(wt), R, C, n, _stream())
    return (wb if want_b else None), (wt if want_t else None)
