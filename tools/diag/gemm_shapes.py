"""per-shape throughput of the GEMM kernels on the shapes of the l_clip step (B=512)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distillclip_amd import ops

def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3

shapes = [(25600, 2304, 768), (25600, 768, 768), (25600, 3072, 768), (25600, 768, 3072), (25600, 768, 2304),
          (39424, 1536, 512), (39424, 512, 512), (39424, 2048, 512), (39424, 512, 2048),
          (39424, 2304, 768), (39424, 768, 768), (39424, 3072, 768), (39424, 768, 3072), (8192, 8192, 8192), (4096, 4096, 4096)]
print('NT  M N K  us  TF/s')
for M, N, K in shapes:
    a = torch.randn(M, K, device='cuda').bfloat16(); b = torch.randn(N, K, device='cuda').bfloat16()
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    t = bench(lambda: ops.gemm_nt(a, b, out=out))
    print(f'NT {M:6d} {N:5d} {K:5d}  {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s   tiles {((M+127)//128)*((N+127)//128)}')
print('TN (wgrad) M P Q')
for M, P, Q in [(25600, 768, 768), (25600, 3072, 768), (25600, 768, 3072), (25600, 2304, 768), (39424, 768, 768), (39424, 3072, 768)]:
    a = torch.randn(M, P, device='cuda').bfloat16(); b = torch.randn(M, Q, device='cuda').bfloat16()
    dw = torch.zeros(P, Q, device='cuda')
    for sp in (2, 4, 8, 16):
        t = bench(lambda: ops.gemm_tn_acc(a, b, dw, sp))
        print(f'TN {M:6d} {P:5d} {Q:5d} splits {sp:2d} {t*1e6:8.1f} us  {2*M*P*Q/t/1e12:7.1f} TF/s')
