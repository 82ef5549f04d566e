#!/usr/bin/env python3
"""isolated time of the two elementwise kernels of the hidden-state loss terms: dclip_feature_mse and dclip_axpy_f32 (with column sums)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distillclip_amd._lib import lib

st = torch.cuda.current_stream().cuda_stream
for M, D in [(25600, 768), (39424, 512), (25600, 512)]:
    s, t, g = torch.randn(M, D, device='cuda'), torch.randn(M, D, device='cuda'), torch.zeros(M, D, device='cuda')
    acc, cs = torch.zeros(1, device='cuda'), torch.zeros(D, device='cuda')
    cp = torch.empty(M, D, device='cuda', dtype=torch.bfloat16)
    fns = {'feature_mse': lambda: lib().dclip_feature_mse(s.data_ptr(), t.data_ptr(), M * D, 0.5, acc.data_ptr(), g.data_ptr(), st),
           'axpy+colsum': lambda: lib().dclip_axpy_f32(g.data_ptr(), s.data_ptr(), cp.data_ptr(), M * D, cs.data_ptr(), D, st)}
    for name, fn in fns.items():
        for _ in range(5):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            fn()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 50 * 1e3
        nbytes = M * D * (16 if name == 'feature_mse' else 14)
        print(f'{name:12s} [{M} x {D}]: {us:7.1f} us  {nbytes / us / 1e6:5.2f} TB/s', flush=True)
