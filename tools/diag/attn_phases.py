"""where a wave of the head-mixing softmax backward spends its cycles per query row (in-kernel stamps, diagnostic launch):
operands -> LDS / dW_w product + row sums / key tiles (two mixes + dS stores) / dW_l product"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distillclip_amd import ops
from distillclip_amd._lib import lib
for B, H, N in [(512, 24, 50), (512, 12, 77)]:
    Np = (N + 7) // 8 * 8
    g = torch.Generator(device='cuda').manual_seed(1)
    dr = torch.randn(B, H, N, Np, device='cuda', generator=g).bfloat16()
    p = torch.softmax(torch.randn(B, H, N, Np, device='cuda', generator=g), -1).bfloat16()
    s = torch.randn(B, H, N, Np, device='cuda', generator=g).bfloat16()
    wl = torch.randn(H, H, device='cuda', generator=g) * 0.2; ww = torch.randn(H, H, device='cuda', generator=g) * 0.2
    dwl = torch.zeros(H, H, device='cuda'); dww = torch.zeros(H, H, device='cuda')
    for _ in range(10): ops.attn_softmax_bwd(dr, p, s, wl, ww, dwl, dww)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.attn_softmax_bwd(dr, p, s, wl, ww, dwl, dww)
    e1.record(); torch.cuda.synchronize()
    buf = torch.zeros(2048 * 4 * 4 * 8, dtype=torch.int64, device='cuda')
    lib().dclip_trace_attn_stamps(buf.data_ptr())
    ops.attn_softmax_bwd(dr, p, s, wl, ww, dwl, dww); torch.cuda.synchronize()
    lib().dclip_trace_attn_stamps(None)
    t = buf.view(-1, 4, 8).cpu().double()
    t = t[t[:, 0, 0] != 0]
    it = t[:, 1:3, :]                                   # iterations 1 and 2 (steady state)
    d = [(it[:, :, k + 1] - it[:, :, k]).flatten().median().item() for k in range(4)]
    nxt = (t[:, 2, 0] - t[:, 1, 0]).median().item()
    print(f'B {B} H {H} N {N}: {e0.elapsed_time(e1) * 100:.1f} us / launch; waves {t.shape[0]}; per row (median cycles): stage {d[0]:.0f}  '
          f'dWw+rowsum {d[1]:.0f}  key tiles {d[2]:.0f}  dWl {d[3]:.0f}  | row to row {nxt:.0f}', flush=True)
