"""GPU parity of the LayerNorm / embedding / attention kernels against torch fp32 on identical (bf16-rounded) inputs."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _randn(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).cuda()


def _close(got, ref, tol, what=''):
    err = (got.float() - ref.float()).abs().max().item()
    den = ref.float().abs().max().item() + 1e-9
    assert err / den < tol, (what, err, den)


@pytest.mark.parametrize('M,D', [(7, 128), (1000, 768), (513, 512), (64, 1024)])
def test_layernorm_fwd_bwd(M, D):
    from distillclip_amd import ops
    x = _randn((M, D), 1, 2.0) + 0.5
    g, b = _randn((D,), 2, 0.1) + 1, _randn((D,), 3, 0.1)
    y, mean, rstd = ops.layernorm_fwd(x, g, b)
    xr = x.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (D,), gr, br, 1e-5)
    _close(y, ref, 5e-3, 'y bf16')
    y32, _, _ = ops.layernorm_fwd(x, g, b, out_dtype=torch.float32)
    _close(y32, ref, 2e-6, 'y f32')
    dy = _randn((M, D), 4).to(torch.bfloat16)
    ref.backward(dy.float())
    acc0 = _randn((M, D), 5)
    acc = acc0.clone()
    dxb = torch.zeros(M, D, dtype=torch.bfloat16, device='cuda')
    dg, db = torch.zeros(D, device='cuda'), torch.zeros(D, device='cuda')
    cs = torch.zeros(D, device='cuda')
    ops.layernorm_bwd(dy, x, g, mean, rstd, acc, dx_bf16=dxb, dgamma=dg, dbeta=db, colsum=cs)
    _close(cs, acc.sum(0), 2e-5, 'colsum of the updated residual gradient')
    _close(acc - acc0, xr.grad, 2e-5, 'dx')
    _close(dxb, acc, 5e-3, 'dx bf16 copy')
    _close(dg, gr.grad, 2e-5, 'dgamma')
    _close(db, br.grad, 2e-5, 'dbeta')


@pytest.mark.parametrize('M,D', [(37, 512), (64, 768), (9, 1024), (5, 128)])
def test_layernorm_on_fp16_rows(M, D):
    """dclip_layernorm_fwd_f16: the frozen teacher's LayerNorms read its fp16 residual stream (the type the reference's `precision: 16`
    autocast keeps it in), compute in f32 and return bf16 (the next GEMM's operand), f32, or fp16 (ln_pre: reference _common.py:14-20,
    :208) — the same arithmetic as the f32 kernel on the widened rows, bit for bit"""
    from distillclip_amd import ops
    x16 = (_randn((M, D), 1, 2.0) + 0.5).to(torch.float16)
    g, b = _randn((D,), 2, 0.1) + 1, _randn((D,), 3, 0.1)
    ref = F.layer_norm(x16.float(), (D,), g, b, 1e-5)
    y, mean, rstd = ops.layernorm_fwd(x16, g, b)
    y32, m32, r32 = ops.layernorm_fwd(x16.float(), g, b)
    assert torch.equal(y, y32) and torch.equal(mean, m32) and torch.equal(rstd, r32)
    _close(y, ref, 5e-3, 'y bf16')
    yf, _, _ = ops.layernorm_fwd(x16, g, b, out_dtype=torch.float32)
    _close(yf, ref, 2e-5, 'y f32')
    yh, _, _ = ops.layernorm_fwd(x16, g, b, out_dtype=torch.float16)
    assert yh.dtype == torch.float16 and torch.equal(yh, yf.to(torch.float16))      # one RNE rounding of the f32 result
    idx = torch.tensor([M - 1, 0, M // 2], dtype=torch.int32, device='cuda')
    yi, _, _ = ops.layernorm_fwd(x16, g, b, row_index=idx)
    assert torch.equal(yi, y[idx.long()])


def test_layernorm_row_index():
    from distillclip_amd import ops
    x = _randn((40, 128), 1)
    g, b = _randn((128,), 2) + 1, _randn((128,), 3)
    idx = torch.tensor([3, 17, 39, 0], dtype=torch.int32, device='cuda')
    y, mean, rstd = ops.layernorm_fwd(x, g, b, row_index=idx, out_dtype=torch.float32)
    _close(y, F.layer_norm(x[idx.long()], (128,), g, b), 2e-6)
    dy = _randn((4, 128), 4)
    acc = torch.zeros_like(x)
    ops.layernorm_bwd(dy, x, g, mean, rstd, acc, row_index=idx)
    xr = x.clone().requires_grad_(True)
    F.layer_norm(xr[idx.long()], (128,), g, b).backward(dy)
    _close(acc, xr.grad, 2e-5)


def _qkv(B, N, H, hd, seed):
    D = H * hd
    return _randn((B * N, 3 * D), seed, 1.0, torch.bfloat16)


def _heads(t, B, N, H, hd):
    return t.float().view(B, N, H, hd).permute(0, 2, 1, 3)


@pytest.mark.parametrize('B,N,H,hd', [(3, 17, 4, 32), (2, 13, 2, 64), (5, 50, 24, 32), (3, 77, 12, 64), (2, 101, 12, 64),
                                      (2, 77, 8, 64)])
def test_attention_products(B, N, H, hd):
    from distillclip_amd import ops
    D = H * hd
    qkv = _qkv(B, N, H, hd, 7)
    q, k, v = (_heads(qkv[:, i * D:(i + 1) * D], B, N, H, hd) for i in range(3))
    scale = hd ** -0.5
    # NT: scores
    s = ops.attn_nt(qkv, 3 * D, qkv[:, D:], 3 * D, B, H, N, hd, alpha=scale)
    ref_s = q @ k.transpose(-1, -2) * scale
    _close(s[..., :N], ref_s, 1e-5, 'scores')
    assert torch.count_nonzero(s[..., N:]) == 0
    # NN: context from an arbitrary bf16 "probability" tensor
    Np = s.shape[-1]
    r = torch.zeros(B, H, N, Np, dtype=torch.bfloat16, device='cuda')
    r[..., :N] = _randn((B, H, N, N), 8, 0.3, torch.bfloat16)
    ctx = torch.zeros(B * N, D, dtype=torch.bfloat16, device='cuda')
    ops.attn_nn(r, qkv[:, 2 * D:], 3 * D, ctx, D, hd)
    ref_ctx = (r[..., :N].float() @ v).permute(0, 2, 1, 3).reshape(B * N, D)
    _close(ctx, ref_ctx, 6e-3, 'ctx')
    # TN: dV = R^T dO
    do = _randn((B * N, D), 9, 1.0, torch.bfloat16)
    dv = torch.zeros(B * N, D, dtype=torch.bfloat16, device='cuda')
    ops.attn_tn(r, do, D, dv, D, hd, alpha=0.5)
    ref_dv = 0.5 * (r[..., :N].float().transpose(-1, -2) @ _heads(do, B, N, H, hd)).permute(0, 2, 1, 3).reshape(B * N, D)
    _close(dv, ref_dv, 6e-3, 'dv')


@pytest.mark.parametrize('B,N,H,mix,causal', [(3, 17, 4, True, False), (2, 13, 2, True, False), (4, 50, 24, True, False),
                                              (3, 77, 12, True, False), (3, 77, 8, False, True), (2, 50, 12, False, False),
                                              (2, 101, 12, False, False), (2, 21, 3, False, True), (3, 128, 1, False, True),
                                              (2, 65, 6, False, False), (1, 50, 16, False, False)])
def test_attention_softmax_stage(B, N, H, mix, causal):
    from distillclip_amd import ops
    Np = (N + 7) // 8 * 8
    s = torch.zeros(B, H, N, Np, device='cuda')
    s[..., :N] = _randn((B, H, N, N), 11, 1.5)
    wl = (torch.eye(H, device='cuda') + _randn((H, H), 12, 0.2)) if mix else None
    ww = (torch.eye(H, device='cuda') + _randn((H, H), 13, 0.2)) if mix else None
    p, r = ops.attn_softmax_fwd(s, wl, ww, causal=causal, save_p=True)

    sr = s[..., :N].clone().requires_grad_(True)
    wlr = wl.clone().requires_grad_(True) if mix else None
    wwr = ww.clone().requires_grad_(True) if mix else None
    a = torch.einsum('gh,bhij->bgij', wlr, sr) if mix else sr
    if causal:
        a = a + torch.full((N, N), float('-inf'), device='cuda').triu_(1)
    pr = a.softmax(-1)
    rr = torch.einsum('gh,bhij->bgij', wwr, pr) if mix else pr
    _close(p[..., :N], pr, 5e-3, 'P')
    _close(r[..., :N], rr, 5e-3, 'R')
    assert torch.count_nonzero(r[..., N:]) == 0 and torch.count_nonzero(p[..., N:]) == 0

    # backward: feed the kernel the bf16 P it saved; compare against autograd of the fp32 graph
    dr = torch.zeros(B, H, N, Np, dtype=torch.bfloat16, device='cuda')
    dr[..., :N] = _randn((B, H, N, N), 14, 1.0, torch.bfloat16)
    rr.backward(dr[..., :N].float())
    dwl = torch.zeros(H, H, device='cuda') if mix else None
    dww = torch.zeros(H, H, device='cuda') if mix else None
    ds = ops.attn_softmax_bwd(dr, p, s, wl, ww, dwl, dww)
    _close(ds[..., :N], sr.grad, 1.5e-2, 'dS')
    assert torch.count_nonzero(ds[..., N:]) == 0
    if mix:
        _close(dww, wwr.grad, 3e-2, 'dWw')
        _close(dwl, wlr.grad, 3e-2, 'dWl')   # bf16 dA / S operands; few positions at the tiny sizes


@pytest.mark.parametrize('B,N,H,hd,causal', [(3, 17, 2, 64, False), (2, 13, 2, 64, True), (5, 50, 12, 64, False),
                                             (3, 77, 8, 64, True), (2, 101, 12, 64, False), (3, 50, 4, 32, False),
                                             (2, 128, 2, 64, True), (1, 1, 2, 64, False), (7, 101, 12, 64, True), (3, 112, 4, 32, False),
                                             (2, 97, 8, 32, True), (3, 33, 4, 64, True), (2, 80, 8, 32, False), (5, 96, 2, 64, True), (9, 65, 4, 32, False)])
def test_attention_fused_forward(B, N, H, hd, causal):
    """fused teacher attention (scores / probabilities never in HBM) vs torch fp32 on the same bf16 q, k, v; every key-tile count 1..8,
    incl. the odd ones whose last tile is a 16-key MFMA step (5: N = 77 / 80, 7: N = 101 / 112 / 97) and the workgroup sizes the launch
    picks for them (round 5), causal and not, more (b, h) problems than one workgroup holds"""
    from distillclip_amd import ops
    D = H * hd
    qkv = _qkv(B, N, H, hd, 31)
    q, k, v = (_heads(qkv[:, i * D:(i + 1) * D], B, N, H, hd) for i in range(3))
    s = q @ k.transpose(-1, -2) * hd ** -0.5
    if causal:
        s = s + torch.full((N, N), float('-inf'), device='cuda').triu_(1)
    ref = (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B * N, D)
    got = ops.attn_fused_fwd(qkv, B, N, H, hd, causal)
    _close(got, ref, 8e-3, 'fused ctx')


def test_embed_scatter_add_with_hot_ids():
    """token-embedding gradient incl. the contended padding / SOT / EOT rows (clip.tokenize layout)"""
    from distillclip_amd import ops, synth
    V, D, B, N = 97, 128, 37, 13
    ids = torch.from_numpy(synth.captions(3, B, N, V, 3, 9)).cuda().reshape(-1)
    dx = _randn((B * N, D), 41)
    tab = torch.ones(V, D, device='cuda')
    ops.embed_scatter_add(ids, dx, tab)
    ref = torch.ones(V, D, device='cuda').index_add_(0, ids, dx)
    _close(tab, ref, 1e-5)
    assert ref[0].abs().sum() > 0 and ref[V - 1].abs().sum() > 0


@pytest.mark.parametrize('B,N,H,hd', [(3, 77, 12, 64), (2, 17, 4, 32), (3, 13, 2, 64), (2, 50, 12, 32), (1, 101, 8, 64), (5, 16, 4, 64),
                                      (2, 128, 12, 64), (1, 1, 2, 64), (3, 50, 24, 32), (2, 101, 24, 32), (9, 77, 12, 64), (70, 50, 24, 32),
                                      (2, 16, 8, 32), (5, 31, 8, 32), (7, 127, 8, 32), (3, 50, 12, 32), (2, 33, 4, 64)])
def test_register_resident_mixed_attention_fwd_bwd(B, N, H, hd):
    """dclip_attn_mix_fwd / _bwd (attention_mix.hip; reference weight_share_model.py:101-125): S, A, P, dR live only in registers
    and both head mixes run on the matrix pipe (f16 operands in the forward -- the precision the reference's fp16 autocast gives
    conv_l / conv_w -- hence 1e-3 on the log-sum-exp rows; bf16 on the gradient side).  Shapes include both shipped students
    (H = 24 / hd = 32 / N = 50 and 101; H = 12 / hd = 64 / N = 77) and a batch that needs more than one persistent round.
    Forward R and the softmax statistics against an fp32 graph on the same bf16 q, k; backward dS, dW_l, dW_w against autograd of
    that graph with dR = dO v^T; and the whole attention (ctx, dq, dk, dv through the unchanged nn / tn products) against the
    unfused three-kernel path on the same inputs."""
    from distillclip_amd import ops
    D = H * hd
    Np = (N + 7) // 8 * 8
    qkv = _randn((B * N, 3 * D), 51, 0.7, torch.bfloat16)
    wl = torch.eye(H, device='cuda') + _randn((H, H), 52, 0.15)
    ww = torch.eye(H, device='cuda') + _randn((H, H), 53, 0.15)
    scale = hd ** -0.5
    Rb, lse = ops.attn_mix_fwd(qkv, B, N, H, hd, wl, ww, scale)           # quad-blocked [B,H,Np/4,N,4]
    R = ops.unblock_scores(Rb)
    q, k, v = (_heads(qkv[:, i * D:(i + 1) * D], B, N, H, hd) for i in range(3))
    sr = ((q @ k.transpose(-1, -2)) * scale).requires_grad_(True)
    wlr, wwr = wl.clone().requires_grad_(True), ww.clone().requires_grad_(True)
    a = torch.einsum('gh,bhij->bgij', wlr, sr)
    rr = torch.einsum('gh,bhij->bgij', wwr, a.softmax(-1))
    _close(R[..., :N], rr, 5e-3, 'R')
    _close(lse, torch.logsumexp(a, -1), 1e-3, 'log-sum-exp rows')
    assert torch.count_nonzero(R[..., N:]) == 0
    # backward of the score stage
    d_ctx = _randn((B * N, D), 54, 1.0, torch.bfloat16)
    do = _heads(d_ctx, B, N, H, hd)
    rr.backward(do @ v.transpose(-1, -2))                               # dR = dO v^T
    dwl, dww = torch.zeros(H, H, device='cuda'), torch.zeros(H, H, device='cuda')
    dSb = ops.attn_mix_bwd(qkv, d_ctx, B, N, H, hd, wl, ww, lse, scale, dwl, dww)
    dS = ops.unblock_scores(dSb)
    _close(dS[..., :N], sr.grad, 1.5e-2, 'dS')                          # bf16 dA / dS (max-norm; 1.2e-2 at H = 8, 4e-3 at H = 24)
    assert torch.count_nonzero(dS[..., N:]) == 0
    _close(dww, wwr.grad, 2e-2, 'dWw')                                  # bf16 operands of the weight-gradient MFMAs
    _close(dwl, wlr.grad, 2e-2, 'dWl')
    # against the unfused kernels on the same inputs (what the towers ran before)
    s3 = ops.attn_nt(qkv, 3 * D, qkv[:, D:], 3 * D, B, H, N, hd, alpha=scale)
    p3, r3 = ops.attn_softmax_fwd(s3, wl, ww, save_p=True)
    _close(R[..., :N], r3[..., :N], 8e-3, 'R vs three kernels')
    dr3 = ops.attn_nt(d_ctx, D, qkv[:, 2 * D:], 3 * D, B, H, N, hd, alpha=1.0, out_dtype=torch.bfloat16)
    dwl3, dww3 = torch.zeros(H, H, device='cuda'), torch.zeros(H, H, device='cuda')
    ds3 = ops.attn_softmax_bwd(dr3, p3, s3, wl, ww, dwl3, dww3)
    if N > 1:                                                           # N = 1: dS is exactly 0 here, bf16 rounding noise there
        _close(dS[..., :N], ds3[..., :N], 1.5e-2, 'dS vs three kernels')
    if N > 1:
        _close(dwl, dwl3, 3e-2, 'dWl vs three kernels')
        _close(dww, dww3, 3e-2, 'dWw vs three kernels')
    # accumulation semantics of the weight gradients (+=) and run-to-run stability of the stored tensors
    # the key / query products read the blocked layout directly: same ctx, dV, dQ, dK as from the row-major copies
    for a_blk, a_row, bm, ldb, alpha in ((Rb, R, qkv[:, 2 * D:], 3 * D, 1.0), (dSb, dS, qkv[:, D:], 3 * D, scale)):
        c1, c2 = (torch.zeros(B * N, D, dtype=torch.bfloat16, device='cuda') for _ in range(2))
        ops.attn_nn(a_blk, bm, ldb, c1, D, hd, alpha)
        ops.attn_nn(a_row.contiguous(), bm, ldb, c2, D, hd, alpha)
        assert torch.equal(c1, c2)
    for a_blk, a_row, bm, ldb, alpha in ((Rb, R, d_ctx, D, 1.0), (dSb, dS, qkv, 3 * D, scale)):
        c1, c2 = (torch.zeros(B * N, D, dtype=torch.bfloat16, device='cuda') for _ in range(2))
        ops.attn_tn(a_blk, bm, ldb, c1, D, hd, alpha)
        ops.attn_tn(a_row.contiguous(), bm, ldb, c2, D, hd, alpha)
        assert torch.equal(c1, c2)
    dS2 = ops.unblock_scores(ops.attn_mix_bwd(qkv, d_ctx, B, N, H, hd, wl, ww, lse, scale, dwl, dww))
    assert torch.equal(dS2, dS)
    _close(dww, 2 * wwr.grad, 2e-2, 'dWw accumulates')
    R2, lse2 = ops.attn_mix_fwd(qkv, B, N, H, hd, wl, ww, scale)
    assert torch.equal(ops.unblock_scores(R2), R) and torch.equal(lse2, lse)


def test_register_resident_mixed_attention_random_shapes():
    """60 random (B, N, H, hd) through the score stage against fp32 autograd, run-to-run equality included (tools/diag/mix_fuzz.py).
    This is the test that found the MFMA-operand hazard of round 3 — an operand register rewritten by the VALU a few instructions
    after the last of four queued MFMAs, H = 8 / hd = 32 only (attention_mix.hip: hw::keep_alive) — which no fixed shape list had."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    torch.cuda.synchronize()
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'diag', 'mix_fuzz.py'), '60', '23'], capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_multi_tensor_cast_transpose():
    """dclip_cast_transpose_bf16_multi (the per-step bf16 weight-cache refresh of a student tower, one launch): every job equals
    the f32 -> bf16 cast and its transpose exactly; more jobs than one launch holds; an odd shape takes the single-tensor kernel"""
    from distillclip_amd import ops
    shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072), (512, 768), (64, 64), (100, 72)] + [(128, 64 * (1 + i % 3)) for i in range(30)]
    ws = [_randn(sh, 70 + i) for i, sh in enumerate(shapes)]
    wb, wt = ops.cast_transpose_multi(ws)
    for w, b, t in zip(ws, wb, wt):
        assert torch.equal(b, w.bfloat16()) and torch.equal(t, w.bfloat16().t().contiguous())
    _, wt2 = ops.cast_transpose_multi(ws[:3], want_b=False)
    assert all(torch.equal(a, b) for a, b in zip(wt2, wt[:3]))


@pytest.mark.parametrize('M,D', [(25600, 512), (39424, 768), (51, 1000), (3, 64), (4100, 1024)])
def test_residual_gradient_add_with_column_sums(M, D):
    """dclip_axpy_f32 (the hidden-state / embedding gradient entering the residual-stream gradient: dst += src, bf16 copy of dst, column sums of
    src = the bias gradient of the linear that wrote the stream).  Round 5: a workgroup owns 256 columns of a row range and issues one atomic per
    column (it was one atomic per element); the add and the copy are exact, the sums within f32 summation-order noise."""
    from distillclip_amd._lib import lib
    torch.manual_seed(M + D)
    dst, src = torch.randn(M, D, device='cuda'), torch.randn(M, D, device='cuda')
    cs = torch.randn(D, device='cuda')
    want_dst, want_cs = dst + src, cs.double() + src.double().sum(0)
    copy = torch.empty(M, D, device='cuda', dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    lib().dclip_axpy_f32(dst.data_ptr(), src.data_ptr(), copy.data_ptr(), M * D, cs.data_ptr(), D, st)
    assert torch.equal(dst, want_dst) and torch.equal(copy, want_dst.bfloat16())
    assert (cs.double() - want_cs).abs().max().item() <= 2e-5 * (M ** 0.5) + 1e-5
    lib().dclip_axpy_f32(dst.data_ptr(), src.data_ptr(), None, M * D, None, D, st)           # no copy, no sums
    assert torch.equal(dst, want_dst + src)
