import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distillclip_amd import ops
for M, N, K in [(25600, 768, 768), (25600, 3072, 768), (25600, 768, 3072), (39424, 2304, 768)]:
    a = torch.randn(M, K, device='cuda').bfloat16(); b = torch.randn(N, K, device='cuda').bfloat16()
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    for _ in range(3): ops.gemm_nt(a, b, out=out)
    torch.cuda.synchronize()
