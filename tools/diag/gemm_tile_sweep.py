#!/usr/bin/env python3
"""Isolated time of every (shape, epilogue) combination the l_clip step launches through dclip_gemm_nt, under the tile-height policy of THIS
process (DCLIP_GEMM320: unset / 1 = the launcher's cycle model, 0 = 256 rows, 2 = 320 rows, 6 = 192 rows; latched at the first launch), three
rotating operand sets per combination.  tools/diag/gemm_tile_sweep.sh runs the four policies and prints the per-combination best."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distillclip_amd import ops

# (M, N, K, epilogue, launches per step)
COMBOS = [(25600, 2304, 768, 'bf16', 18), (25600, 3072, 768, 'qgelu', 12), (25600, 768, 3072, 'res16', 12), (39424, 2048, 512, 'qgelu', 12),
          (39424, 512, 2048, 'res16', 12), (39424, 3072, 768, 'gelu_save', 4), (25600, 3072, 768, 'gelu_save', 6), (25600, 3072, 768, 'mulaux', 6),
          (39424, 3072, 768, 'mulaux', 4), (39424, 1536, 512, 'bf16', 12), (39424, 768, 3072, 'res', 4), (25600, 768, 3072, 'res', 6),
          (39424, 768, 3072, 'bf16', 4), (25600, 768, 3072, 'bf16', 6), (39424, 2304, 768, 'bf16', 4), (25600, 768, 768, 'res16', 12),
          (39424, 768, 2304, 'bf16', 4), (39424, 512, 512, 'res16', 12), (25600, 768, 2304, 'bf16', 6), (39424, 768, 768, 'res', 4),
          (25600, 768, 768, 'res', 6), (39424, 768, 768, 'bf16', 4), (25600, 768, 768, 'bf16', 6)]
NSET = 3
out = {}
for M, N, K, kind, per_step in COMBOS:
    sets = []
    for s in range(NSET):
        a = torch.randn(M, K, device='cuda').bfloat16()
        b = (torch.randn(N, K, device='cuda') * 0.05).bfloat16()
        d = dict(a=a, b=b, bias=torch.randn(N, device='cuda'))
        if kind == 'res':
            d['res'] = torch.randn(M, N, device='cuda')
        elif kind == 'res16':
            d['res'] = torch.randn(M, N, device='cuda').to(torch.float16)
        elif kind in ('gelu_save', 'mulaux'):
            d['aux'] = torch.randint(0, 255, (M, N), device='cuda', dtype=torch.uint8)
            d['cs'] = torch.zeros(N, device='cuda')
        sets.append(d)

    def run(d):
        if kind in ('res', 'res16'):
            ops.gemm_nt(d['a'], d['b'], bias=d['bias'], residual=d['res'], out=d['res'])
        elif kind == 'qgelu':
            ops.gemm_nt(d['a'], d['b'], bias=d['bias'], act='quickgelu')
        elif kind == 'gelu_save':
            ops.gemm_nt(d['a'], d['b'], bias=d['bias'], act='gelu_save', aux_out=d['aux'])
        elif kind == 'mulaux':
            ops.gemm_nt(d['a'], d['b'], act='mulaux', aux_in=d['aux'], colsum=d['cs'])
        else:
            ops.gemm_nt(d['a'], d['b'])
    for i in range(6):
        run(sets[i % NSET])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for i in range(n):
        run(sets[i % NSET])
    e1.record()
    torch.cuda.synchronize()
    out[f'{M}x{N}x{K}:{kind}'] = dict(us=e0.elapsed_time(e1) / n * 1e3, per_step=per_step)
    del sets
    torch.cuda.empty_cache()
print(json.dumps(dict(mode=os.environ.get('DCLIP_GEMM320', 'model'), combos=out)))
