import sys, os
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/distillclip_amd') else os.environ.get('GRAFT_REPO_ROOT','.'))
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
for M, P, Q in [(25600, 3072, 768), (25600, 768, 3072), (25600, 2304, 768), (39424, 3072, 768), (39424, 2048, 512)]:
    a = torch.randn(M, P, device='cuda').bfloat16(); b = torch.randn(M, Q, device='cuda').bfloat16()
    dw = torch.zeros(P, Q, device='cuda')
    t = bench(lambda: ops.gemm_tn_acc(a, b, dw, 4))
    print(f'TN {M:6d} {P:5d} {Q:5d} {t*1e6:8.1f} us  {2*M*P*Q/t/1e12:7.1f} TF/s', flush=True)
