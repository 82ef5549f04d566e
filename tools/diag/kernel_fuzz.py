"""random shapes through gemm_nt (all epilogues), the attention products and the teacher's fused attention against torch fp32 on the same
bf16 inputs (run-to-run equality included), plus wgrad and LayerNorm forward: python tools/diag/kernel_fuzz.py [cases] [seed]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distillclip_amd import ops
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad = 0
rel = lambda x, y: ((x.float() - y.float()).abs().max() / (y.float().abs().max() + 1e-9)).item()


def rnd(shape, seed, scale=1.0, dtype=torch.bfloat16):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).cuda()


for t in range(cases):
    # ---- gemm_nt
    M = rng.choice([1024, 2048, 4096, 12800, 12810, 25600, 39424, rng.randrange(1024, 30000)])
    N = rng.choice([256, 512, 520, 768, 1536, 2304, 3072, 8 * rng.randrange(32, 400)])
    K = 64 * rng.randrange(1, 49)
    kind = rng.choice(['plain', 'bias', 'qgelu', 'gelu_save', 'mulaux_cs', 'f32res', 'f32', 'inplace', 'f16inplace'])
    a, b = rnd((M, K), 10 * t + 1), rnd((N, K), 10 * t + 2, 0.2)
    ref = a.float() @ b.float().t()
    bias = torch.randn(N, device='cuda')
    tol16, tol32 = 8e-3, 1e-5 * K ** 0.5 + 1e-5
    errs = {}
    if kind == 'plain':
        o = ops.gemm_nt(a, b); o2 = ops.gemm_nt(a, b)
        errs['o'] = (rel(o, ref), tol16); same = torch.equal(o, o2)
    elif kind == 'bias':
        o = ops.gemm_nt(a, b, bias=bias); o2 = ops.gemm_nt(a, b, bias=bias)
        errs['o'] = (rel(o, ref + bias), tol16); same = torch.equal(o, o2)
    elif kind == 'qgelu':
        z = ref + bias
        o = ops.gemm_nt(a, b, bias=bias, act='quickgelu'); o2 = ops.gemm_nt(a, b, bias=bias, act='quickgelu')
        errs['o'] = (rel(o, z * torch.sigmoid(1.702 * z)), tol16); same = torch.equal(o, o2)
    elif kind == 'gelu_save':
        aux = torch.empty(M, N, dtype=torch.uint8, device='cuda')            # gelu' as 8-bit fixed point (include/dclip.h)
        zz = (ref + bias).requires_grad_(True)
        g = torch.nn.functional.gelu(zz); g.sum().backward()
        o = ops.gemm_nt(a, b, bias=bias, act='gelu_save', aux_out=aux)
        errs['o'] = (rel(o, g.detach()), tol16)
        errs['aux'] = ((aux.float() * ops.DG_STEP + ops.DG_LO - zz.grad).abs().max().item(), 0.5 * ops.DG_STEP + 1e-5)
        aux2 = torch.empty_like(aux); o2 = ops.gemm_nt(a, b, bias=bias, act='gelu_save', aux_out=aux2); same = torch.equal(o, o2) and torch.equal(aux, aux2)
    elif kind == 'mulaux_cs':
        z = torch.randint(0, 256, (M, N), dtype=torch.uint8, generator=torch.Generator().manual_seed(10 * t + 3)).cuda()
        cs = torch.zeros(N, device='cuda')
        o = ops.gemm_nt(a, b, act='mulaux', aux_in=z, colsum=cs)
        want = ref * (z.float() * ops.DG_STEP + ops.DG_LO)
        errs['o'] = (rel(o, want), tol16); errs['cs'] = (rel(cs, want.sum(0)), 4e-3)
        o2 = ops.gemm_nt(a, b, act='mulaux', aux_in=z); same = torch.equal(o, o2)
    elif kind == 'f32res':
        res = torch.randn(M, N, device='cuda')
        o = ops.gemm_nt(a, b, bias=bias, residual=res, out_dtype=torch.float32); o2 = ops.gemm_nt(a, b, bias=bias, residual=res, out_dtype=torch.float32)
        errs['o'] = (rel(o, ref + bias + res), max(tol32, 1e-4)); same = torch.equal(o, o2)
    elif kind == 'f32':
        o = ops.gemm_nt(a, b, out_dtype=torch.float32); o2 = ops.gemm_nt(a, b, out_dtype=torch.float32)
        errs['o'] = (rel(o, ref), tol32); same = torch.equal(o, o2)
    elif kind == 'f16inplace':                                                  # the frozen teacher's fp16 residual stream
        res = torch.randn(M, N, device='cuda').to(torch.float16); x = res.clone(); x2 = res.clone()
        ops.gemm_nt(a, b, bias=bias, residual=x, out=x); ops.gemm_nt(a, b, bias=bias, residual=x2, out=x2)
        errs['o'] = (rel(x, ref + bias + res.float()), max(tol32, 1.2e-3)); same = torch.equal(x, x2)
    else:
        res = torch.randn(M, N, device='cuda'); x = res.clone(); x2 = res.clone()
        ops.gemm_nt(a, b, bias=bias, residual=x, out=x); ops.gemm_nt(a, b, bias=bias, residual=x2, out=x2)
        errs['o'] = (rel(x, ref + bias + res), max(tol32, 1e-4)); same = torch.equal(x, x2)
    ok = same and all(e < tol for e, tol in errs.values())
    if not ok:
        bad += 1
        print('FAIL gemm', (M, N, K), kind, errs, 'run-to-run equal', same, flush=True)
    # ---- attention products + teacher attention
    H, hd = rng.choice([(2, 64), (4, 32), (8, 32), (8, 64), (12, 64), (24, 32)])
    Nq = rng.choice([1, 7, 16, 33, 50, 64, 77, 101, 128, rng.randint(1, 128)])
    B = rng.choice([1, 3, 9, 40])
    D = H * hd
    Np = (Nq + 7) // 8 * 8
    qkv = rnd((B * Nq, 3 * D), 10 * t + 5, 0.7)
    heads = lambda x: x.float().view(B, Nq, H, hd).permute(0, 2, 1, 3)
    q, k, v = heads(qkv[:, :D]), heads(qkv[:, D:2 * D]), heads(qkv[:, 2 * D:])
    for causal in (False, True):
        s = (q @ k.transpose(-1, -2)) * hd ** -0.5
        if causal:
            s = s.masked_fill(torch.triu(torch.ones(Nq, Nq, device='cuda', dtype=torch.bool), 1), float('-inf'))
        want = (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B * Nq, D)
        got = ops.attn_fused_fwd(qkv, B, Nq, H, hd, causal)
        e = rel(got, want)
        if not (e < 8e-3 and torch.equal(got, ops.attn_fused_fwd(qkv, B, Nq, H, hd, causal))):
            bad += 1
            print('FAIL fused', (B, Nq, H, hd, causal), e, flush=True)
    A = torch.zeros(B, H, Nq, Np, device='cuda')
    A[..., :Nq] = torch.randn(B, H, Nq, Nq, device='cuda')
    Ab = A.bfloat16()
    Ablk = Ab.view(B, H, Nq, Np // 4, 4).permute(0, 1, 3, 2, 4).contiguous()
    for name, fn, want in (('nn', ops.attn_nn, (Ab.float()[..., :Nq] @ v)), ('tn', ops.attn_tn, (Ab.float()[..., :Nq].transpose(-1, -2) @ v))):
        want = want.permute(0, 2, 1, 3).reshape(B * Nq, D)
        for src in (Ab, Ablk):
            out = torch.zeros(B * Nq, D, dtype=torch.bfloat16, device='cuda')
            fn(src, qkv[:, 2 * D:], 3 * D, out, D, hd, 1.0)
            e = rel(out, want)
            if not e < 8e-3:
                bad += 1
                print('FAIL', name, (B, Nq, H, hd), 'blocked' if src is Ablk else 'row-major', e, flush=True)
    # ---- wgrad (gemm_tn_acc), LayerNorm forward / backward
    Mw = rng.choice([1024, 4096, 25600, 51200, 78848, 64 * rng.randrange(16, 900)])
    P, Q = rng.choice([256, 512, 768, 2304, 3072]), rng.choice([256, 512, 768, 2304, 3072])
    xa, xb = rnd((Mw, P), 10 * t + 6, 0.5), rnd((Mw, Q), 10 * t + 7, 0.5)
    want = xa.float().t() @ xb.float()
    dws = []
    for _ in range(2):
        dw = torch.zeros(P, Q, device='cuda')
        ops.gemm_tn_acc(xa, xb, dw, splits=rng.choice([2, 4, 8]))
        dws.append(dw)
    e = rel(dws[0], want)
    if not e < 2e-4:
        bad += 1
        print('FAIL gemm_tn', (Mw, P, Q), e, flush=True)
    Dl = rng.choice([512, 768])
    Ml = rng.choice([512, 25600, 39424, rng.randrange(100, 20000)])
    x = torch.randn(Ml, Dl, device='cuda'); gm = torch.randn(Dl, device='cuda'); bt = torch.randn(Dl, device='cuda')
    y, mean, rstd = ops.layernorm_fwd(x, gm, bt)
    yr = torch.nn.functional.layer_norm(x, (Dl,), gm, bt)
    if not rel(y, yr) < 6e-3:
        bad += 1
        print('FAIL ln_fwd', (Ml, Dl), rel(y, yr), flush=True)
print('cases', cases, 'failed', bad)
sys.exit(1 if bad else 0)
