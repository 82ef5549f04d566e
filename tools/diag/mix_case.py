"""one shape through dclip_attn_mix_bwd against fp32 autograd, with the pattern of the wrong dS entries: python tools/diag/mix_case.py B N H hd"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distillclip_amd import ops
B, N, H, hd = (int(x) for x in sys.argv[1:5])
D = H * hd
g = torch.Generator(device='cpu').manual_seed(5)
qkv = (torch.randn(B * N, 3 * D, generator=g) * 0.7).bfloat16().cuda()
dctx = torch.randn(B * N, D, generator=g).bfloat16().cuda()
wl = (torch.eye(H) + 0.15 * torch.randn(H, H, generator=g)).cuda()
ww = (torch.eye(H) + 0.15 * torch.randn(H, H, generator=g)).cuda()
scale = hd ** -0.5
Rb, lse = ops.attn_mix_fwd(qkv, B, N, H, hd, wl, ww, scale)
heads = lambda x: x.float().view(B, N, H, hd).permute(0, 2, 1, 3)
q, k, v = heads(qkv[:, :D]), heads(qkv[:, D:2 * D]), heads(qkv[:, 2 * D:])
sr = ((q @ k.transpose(-1, -2)) * scale).requires_grad_(True)
wlr, wwr = wl.clone().requires_grad_(True), ww.clone().requires_grad_(True)
a = torch.einsum('gh,bhij->bgij', wlr, sr)
p = a.softmax(-1); p.retain_grad()
rr = torch.einsum('gh,bhij->bgij', wwr, p)
rr.backward(heads(dctx) @ v.transpose(-1, -2))
outs = []
for _ in range(2):
    dwl, dww = torch.zeros(H, H, device='cuda'), torch.zeros(H, H, device='cuda')
    dS = ops.unblock_scores(ops.attn_mix_bwd(qkv, dctx, B, N, H, hd, wl, ww, lse, scale, dwl, dww)).float()[..., :N]
    outs.append((dS, dwl.clone(), dww.clone()))
print('run-to-run equal', all(torch.equal(x, y) for x, y in zip(*outs)))
dS, dwl, dww = outs[0]
err = (dS - sr.grad).abs()
print('dS rel', (err.max() / sr.grad.abs().max()).item(), 'dWl rel', ((dwl - wlr.grad).abs().max() / wlr.grad.abs().max()).item(),
      'dWw rel', ((dww - wwr.grad).abs().max() / wwr.grad.abs().max()).item())
bad = (err > 0.02 * sr.grad.abs().max()).nonzero()
print('bad', bad.shape[0], 'of', err.numel())
if bad.shape[0]:
    print('by head', sorted(collections.Counter(bad[:, 1].tolist()).items()))
    print('by query', sorted(collections.Counter(bad[:, 2].tolist()).items())[:40])
    print('by key', sorted(collections.Counter(bad[:, 3].tolist()).items())[:40])
    # is the wrong dS what a different weight orientation / head permutation would give?
    dA = p.grad * p.detach() - p.detach() * (p.grad * p.detach()).sum(-1, keepdim=True)
    for name, cand in (('W_l (not transposed)', torch.einsum('hg,bgij->bhij', wl, dA)), ('W_l^T (correct)', torch.einsum('gh,bgij->bhij', wl, dA))):
        print('   dS vs', name, ((dS - cand).abs().max() / cand.abs().max()).item())
