"""teacher fused attention, isolated: us per launch at the shipped shapes (B = 512): ViT-B/32 224 px (N = 50), 336 px (N = 101), text (N = 77, causal)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distillclip_amd import ops
for B, N, H, hd, causal in ((512, 50, 12, 64, False), (512, 101, 12, 64, False), (512, 77, 8, 64, True), (512, 50, 24, 32, False), (512, 101, 24, 32, False)):
    D = H * hd
    qkvs = [torch.randn(B * N, 3 * D, device='cuda').to(torch.bfloat16) for _ in range(4)]
    for q in qkvs: ops.attn_fused_fwd(q, B, N, H, hd, causal)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(40): ops.attn_fused_fwd(qkvs[i % 4], B, N, H, hd, causal)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 40 * 1e3
    print(f'attn_fused_fwd B={B} N={N} H={H} hd={hd} causal={causal}: {us:.1f} us  ({8.0 * B * N * D / us / 1e6:.2f} TB/s algorithmic)')
