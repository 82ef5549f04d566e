"""global-negative loss at 8 ranks x 512 pairs: every rank evaluating the whole [4096, 4096] matrix vs its own [512, 4096] row block"""
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
    return a.elapsed_time(b) / n

w = {'out_l1': 1 / 3, 'out_cos': 1 / 3, 'cos_diff': 0.1 / 3}
for Bg in (512, 1024, 2048, 4096):
    e = [torch.randn(Bg, 512, device='cuda') for _ in range(4)]
    t_full = bench(lambda: ops.distill_loss(*e, weights=w))
    t_rows = bench(lambda: ops.distill_loss(*e, weights=w, row0=0, rows=512))
    print(f'gathered batch {Bg:5d}: whole matrix {t_full:7.3f} ms   own 512-row block {t_rows:7.3f} ms')
