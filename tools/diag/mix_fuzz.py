"""random shapes through dclip_attn_mix_fwd / _bwd against an fp32 graph (beyond the parametrised test): python tools/diag/mix_fuzz.py [cases] [seed]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distillclip_amd import ops
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
from distillclip_amd._lib import lib
shapes = [hs for hs in [(2, 32), (2, 64), (4, 32), (4, 64), (8, 32), (8, 64), (12, 32), (12, 64), (24, 32)] if lib().dclip_attn_mix_supported(hs[0], 50, hs[1])]
worst = {}
bad = 0
for t in range(cases):
    H, hd = rng.choice(shapes)
    N = rng.choice([1, 2, 3, 4, 5, 15, 16, 17, 31, 33, 49, 50, 63, 64, 65, 77, 80, 96, 101, 127, 128, rng.randint(1, 128)])
    B = rng.choice([1, 2, 3, 5, 17, 64]) if N <= 64 else rng.choice([1, 2, 7])
    D = H * hd
    g = torch.Generator(device='cpu').manual_seed(1000 + t)
    qkv = (torch.randn(B * N, 3 * D, generator=g) * 0.7).bfloat16().cuda()
    dctx = torch.randn(B * N, D, generator=g).bfloat16().cuda()
    wl = (torch.eye(H) + 0.15 * torch.randn(H, H, generator=g)).cuda()
    ww = (torch.eye(H) + 0.15 * torch.randn(H, H, generator=g)).cuda()
    scale = hd ** -0.5
    Rb, lse = ops.attn_mix_fwd(qkv, B, N, H, hd, wl, ww, scale)
    R = ops.unblock_scores(Rb).float()
    heads = lambda x: x.float().view(B, N, H, hd).permute(0, 2, 1, 3)
    q, k, v = heads(qkv[:, :D]), heads(qkv[:, D:2 * D]), heads(qkv[:, 2 * D:])
    sr = ((q @ k.transpose(-1, -2)) * scale).requires_grad_(True)
    wlr, wwr = wl.clone().requires_grad_(True), ww.clone().requires_grad_(True)
    a = torch.einsum('gh,bhij->bgij', wlr, sr)
    rr = torch.einsum('gh,bhij->bgij', wwr, a.softmax(-1))
    rr.backward(heads(dctx) @ v.transpose(-1, -2))
    dwl, dww = torch.zeros(H, H, device='cuda'), torch.zeros(H, H, device='cuda')
    dSb = ops.attn_mix_bwd(qkv, dctx, B, N, H, hd, wl, ww, lse, scale, dwl, dww)
    dS = ops.unblock_scores(dSb).float()
    dwl2, dww2 = torch.zeros(H, H, device='cuda'), torch.zeros(H, H, device='cuda')
    dSb2 = ops.attn_mix_bwd(qkv, dctx, B, N, H, hd, wl, ww, lse, scale, dwl2, dww2)
    rel = lambda x, y: ((x - y).abs().max() / (y.abs().max() + 1e-9)).item()
    errs = dict(R=rel(R[..., :N], rr.detach()), lse=rel(lse, torch.logsumexp(a.detach(), -1)), dS=rel(dS[..., :N], sr.grad) if N > 1 else 0.0,
                dWl=rel(dwl, wlr.grad) if N > 1 else 0.0, dWw=rel(dww, wwr.grad))
    ok = errs['R'] < 5e-3 and errs['lse'] < 1e-3 and errs['dS'] < 1.5e-2 and errs['dWl'] < 2e-2 and errs['dWw'] < 2e-2
    ok = ok and torch.equal(dSb, dSb2) and torch.equal(dwl, dwl2) and torch.equal(dww, dww2)
    ok = ok and torch.count_nonzero(R[..., N:]) == 0 and torch.count_nonzero(dS[..., N:]) == 0 and bool(torch.isfinite(R).all()) and bool(torch.isfinite(dS).all())
    for kk, vv in errs.items():
        worst[kk] = max(worst.get(kk, 0.0), vv)
    if not ok:
        bad += 1
        print('FAIL', (B, N, H, hd), errs, flush=True)
print('cases', cases, 'failed', bad, 'worst', {k: round(v, 5) for k, v in worst.items()})
sys.exit(1 if bad else 0)
