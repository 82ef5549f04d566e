"""calibration: the library GEMM (torch.matmul -> hipBLASLt / rocBLAS, bf16, f32 accumulate) on the step's shapes, sustained,
next to dclip_gemm_nt.  Not used by the product: it only tells how far the hand-written kernel is from the vendor's."""
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

shapes = [(25600, 2304, 768), (25600, 768, 768), (25600, 3072, 768), (25600, 768, 3072), (39424, 1536, 512), (39424, 512, 512),
          (39424, 2048, 512), (39424, 512, 2048), (39424, 2304, 768), (39424, 768, 768), (39424, 3072, 768), (39424, 768, 3072),
          (4096, 4096, 4096), (8192, 8192, 8192)]
print('M N K | ours us TF/s | library us TF/s')
for M, N, K in shapes:
    a = torch.randn(M, K, device='cuda').bfloat16(); b = (torch.randn(N, K, device='cuda') * 0.05).bfloat16()
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    bt = b.t()
    t1 = bench(lambda: ops.gemm_nt(a, b, out=out))
    t2 = bench(lambda: torch.matmul(a, bt, out=out))
    f = 2 * M * N * K
    print(f'{M:6d} {N:5d} {K:5d} | {t1*1e6:7.1f} {f/t1/1e12:7.1f} | {t2*1e6:7.1f} {f/t2/1e12:7.1f}', flush=True)
