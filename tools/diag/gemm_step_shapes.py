"""sustained in-isolation time of the step's GEMM shapes (plain / f32+residual epilogues), 256-row vs 320-row tiles are chosen by
the launcher (DCLIP_GEMM320=0 to compare).  python tools/diag/gemm_step_shapes.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distillclip_amd import ops
shapes = [(25600, 768, 768, 'res16'), (25600, 768, 3072, 'res16'), (39424, 512, 512, 'res16'), (39424, 512, 2048, 'res16'),
          (25600, 768, 3072, 'res'), (25600, 768, 768, 'res'), (25600, 2304, 768, 'bf16'), (25600, 3072, 768, 'qgelu'),
          (39424, 512, 2048, 'res'), (39424, 512, 512, 'res'), (39424, 1536, 512, 'bf16'), (39424, 2048, 512, 'qgelu'),
          (25600, 768, 3072, 'bf16'), (25600, 768, 2304, 'bf16'), (39424, 768, 3072, 'res'), (39424, 3072, 768, 'bf16')]
for M, N, K, kind in shapes:
    a = torch.randn(M, K, device='cuda').bfloat16(); b = (torch.randn(N, K, device='cuda') * 0.05).bfloat16()
    bias = torch.randn(N, device='cuda'); res = torch.randn(M, N, device='cuda'); out = torch.empty(M, N, device='cuda')
    res16 = res.to(torch.float16)           # the frozen teacher's fp16 residual stream, in place
    def run():
        if kind == 'res': ops.gemm_nt(a, b, bias=bias, residual=res, out=out)
        elif kind == 'res16': ops.gemm_nt(a, b, bias=bias, residual=res16, out=res16)
        elif kind == 'qgelu': ops.gemm_nt(a, b, bias=bias, act='quickgelu')
        else: ops.gemm_nt(a, b)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f'{M:6d} {N:5d} {K:5d} {kind:6s} {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s', flush=True)
