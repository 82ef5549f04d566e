"""where a 256^2 wgrad (gemm_tn256) workgroup spends its cycles: prologue / contraction loop / atomic epilogue"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distillclip_amd import ops
from distillclip_amd._lib import lib
for M, P, Q in [(51200, 3072, 768), (51200, 768, 3072), (51200, 2304, 768), (51200, 768, 768), (78848, 3072, 768), (78848, 768, 768)]:
    a = torch.randn(M, P, device='cuda').bfloat16(); b = torch.randn(M, Q, device='cuda').bfloat16()
    dw = torch.zeros(P, Q, device='cuda')
    for _ in range(10): ops.gemm_tn_acc(a, b, dw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.gemm_tn_acc(a, b, dw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    buf = torch.zeros(6 * 4096, dtype=torch.int64, device='cuda')
    lib().dclip_trace_gemm_stamps(buf.data_ptr())
    ops.gemm_tn_acc(a, b, dw); torch.cuda.synchronize()
    lib().dclip_trace_gemm_stamps(None)
    s = buf.view(-1, 6).cpu(); s = s[s[:, 0] != 0].double()
    pro, main, epi = (s[:, 1] - s[:, 0]), (s[:, 2] - s[:, 1]), (s[:, 3] - s[:, 2])
    clk = ((s[:, 3] - s[:, 0]) / (s[:, 5] - s[:, 4]).clamp(min=1) * 0.1).median().item()
    print(f'M {M} P {P} Q {Q}: {us:7.1f} us = {2.0 * M * P * Q / us / 1e6:6.0f} TFLOP/s | wgs {s.shape[0]} clock {clk:4.2f} GHz | median cycles: prologue {pro.median().item():6.0f} '
          f'main {main.median().item():8.0f} epilogue {epi.median().item():7.0f}', flush=True)
