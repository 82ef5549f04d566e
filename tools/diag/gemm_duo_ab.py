"""The l_clip step's 22 gemm_nt (shape, epilogue) combinations with their launch counts per step, timed back to back in isolation
(each combination rotates over 3 operand / output sets so that a launch does not find its own previous output in the caches), and the
count-weighted sum = the step's gemm_nt time under this selection.  The selection knobs are latched per process: run once per setting,
    DCLIP_GEMM_DUO=0 python tools/diag/gemm_duo_ab.py ; DCLIP_GEMM_DUO=2 python tools/diag/gemm_duo_ab.py [filter]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distillclip_amd import ops
COMBOS = [(25600, 768, 3072, 'res', 18), (25600, 2304, 768, 'bf16', 18), (25600, 3072, 768, 'qgelu', 12), (39424, 2048, 512, 'qgelu', 12),
          (39424, 512, 2048, 'res', 12), (25600, 768, 768, 'res', 18), (39424, 3072, 768, 'gelu_save', 4), (25600, 3072, 768, 'gelu_save', 6),
          (39424, 3072, 768, 'mulaux', 4), (25600, 3072, 768, 'mulaux', 6), (39424, 1536, 512, 'bf16', 12), (39424, 768, 3072, 'res', 4),
          (39424, 512, 512, 'res', 12), (25600, 768, 3072, 'bf16', 6), (39424, 2304, 768, 'bf16', 4), (39424, 768, 3072, 'bf16', 4),
          (39424, 768, 2304, 'bf16', 4), (25600, 768, 2304, 'bf16', 6), (39424, 768, 768, 'res', 4), (25600, 768, 3072, 'f32', 2),
          (39424, 768, 768, 'bf16', 4), (25600, 768, 768, 'bf16', 6)]
flt = sys.argv[1] if len(sys.argv) > 1 else ''
NS = 3
tot_ms, tot_fl = 0.0, 0.0
print(f'DUO={os.environ.get("DCLIP_GEMM_DUO", "0")} MI={os.environ.get("DCLIP_DUO_MI", "auto")} PRIO={os.environ.get("DCLIP_DUO_PRIO", "1")}')
for M, N, K, kind, cnt in COMBOS:
    if flt and flt not in f'{M}x{N}x{K}:{kind}':
        continue
    sets = []
    for s in range(NS):
        a = torch.randn(M, K, device='cuda').bfloat16(); b = (torch.randn(N, K, device='cuda') * 0.05).bfloat16()
        sets.append(dict(a=a, b=b, bias=torch.randn(N, device='cuda'), res=torch.randn(M, N, device='cuda'),
                         out32=torch.empty(M, N, device='cuda'), out16=torch.empty(M, N, device='cuda', dtype=torch.bfloat16),
                         aux=torch.randn(M, N, device='cuda').bfloat16(), cs=torch.zeros(N, device='cuda')))
    def run(d):
        if kind == 'res': ops.gemm_nt(d['a'], d['b'], bias=d['bias'], residual=d['res'], out=d['res'])
        elif kind == 'f32': ops.gemm_nt(d['a'], d['b'], bias=d['bias'], out=d['out32'])
        elif kind == 'qgelu': ops.gemm_nt(d['a'], d['b'], bias=d['bias'], act='quickgelu', out=d['out16'])
        elif kind == 'gelu_save': ops.gemm_nt(d['a'], d['b'], bias=d['bias'], act='gelu_save', aux_out=d['aux'], out=d['out16'])
        elif kind == 'mulaux': ops.gemm_nt(d['a'], d['b'], act='mulaux', aux_in=d['aux'], colsum=d['cs'], out=d['out16'])
        else: ops.gemm_nt(d['a'], d['b'], out=d['out16'])
    for i in range(6): run(sets[i % NS])
    torch.cuda.synchronize()
    n = 30
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): run(sets[i % NS])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    tot_ms += us * cnt * 1e-3; tot_fl += 2.0 * M * N * K * cnt
    print(f'{M:6d} {N:5d} {K:5d} {kind:9s} x{cnt:2d} {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s', flush=True)
    del sets
print(f'weighted gemm_nt: {tot_ms:.3f} ms / step, {tot_fl / tot_ms / 1e9:.1f} TF/s = {tot_fl / tot_ms / 1e9 / 2500:.4f} of 2.5 PF')
