"""LayerNorm forward / backward achieved HBM rate on the step's shapes (sustained)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distillclip_amd import ops

def bench(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3

for M, D in [(25600, 768), (39424, 768), (39424, 512)]:
    x = torch.randn(M, D, device='cuda'); g = torch.randn(D, device='cuda'); be = torch.randn(D, device='cuda')
    y, mean, rstd = ops.layernorm_fwd(x, g, be)
    dy = torch.randn(M, D, device='cuda').bfloat16()
    dx = torch.zeros(M, D, device='cuda'); dxb = torch.empty(M, D, device='cuda', dtype=torch.bfloat16)
    dg = torch.zeros(D, device='cuda'); db = torch.zeros(D, device='cuda'); cs = torch.zeros(D, device='cuda')
    t = bench(lambda: ops.layernorm_fwd(x, g, be))
    print(f'fwd  {M:6d} {D:4d} {t*1e6:7.1f} us  {M*D*6/t/1e12:5.2f} TB/s')
    t = bench(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dx, dgamma=dg, dbeta=db))
    print(f'bwd  {M:6d} {D:4d} {t*1e6:7.1f} us  {M*D*14/t/1e12:5.2f} TB/s  (x 4 + dy 2 + dx 4+4)')
    t = bench(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dx, dx_bf16=dxb, dgamma=dg, dbeta=db, colsum=cs))
    print(f'bwd+ {M:6d} {D:4d} {t*1e6:7.1f} us  {M*D*16/t/1e12:5.2f} TB/s  (+ bf16 copy 2)')
    t = bench(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dx, dx_bf16=dxb))
    print(f'bwd0 {M:6d} {D:4d} {t*1e6:7.1f} us  {M*D*16/t/1e12:5.2f} TB/s  (no dgamma / dbeta / column sums: the kernel without its closing atomics)')
