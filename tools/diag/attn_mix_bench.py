"""register-resident mixed attention (attention_mix.hip) against the unfused score kernels, sustained, step shapes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distillclip_amd import ops
from distillclip_amd._lib import lib

def bench(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for B, N, H, hd in [(512, 77, 12, 64), (512, 50, 12, 64), (1024, 77, 12, 64)] + ([(512, 50, 24, 32)] if lib().dclip_attn_mix_supported(24, 50, 32) else []):
    D = H * hd
    qkv = (torch.randn(B * N, 3 * D, device='cuda') * 0.7).bfloat16()
    dctx = torch.randn(B * N, D, device='cuda').bfloat16()
    wl = torch.eye(H, device='cuda') + 0.1 * torch.randn(H, H, device='cuda')
    ww = torch.eye(H, device='cuda') + 0.1 * torch.randn(H, H, device='cuda')
    scale = hd ** -0.5
    dwl, dww = torch.zeros(H, H, device='cuda'), torch.zeros(H, H, device='cuda')
    R, lse = ops.attn_mix_fwd(qkv, B, N, H, hd, wl, ww, scale)
    t_f = bench(lambda: ops.attn_mix_fwd(qkv, B, N, H, hd, wl, ww, scale))
    t_b = bench(lambda: ops.attn_mix_bwd(qkv, dctx, B, N, H, hd, wl, ww, lse, scale, dwl, dww))
    s3 = ops.attn_nt(qkv, 3 * D, qkv[:, D:], 3 * D, B, H, N, hd, alpha=scale)
    p3, r3 = ops.attn_softmax_fwd(s3, wl, ww, save_p=True)
    dr3 = ops.attn_nt(dctx, D, qkv[:, 2 * D:], 3 * D, B, H, N, hd, alpha=1.0, out_dtype=torch.bfloat16)
    o_f = bench(lambda: (ops.attn_nt(qkv, 3 * D, qkv[:, D:], 3 * D, B, H, N, hd, alpha=scale), ops.attn_softmax_fwd(s3, wl, ww, save_p=True)))
    o_b = bench(lambda: (ops.attn_nt(dctx, D, qkv[:, 2 * D:], 3 * D, B, H, N, hd, alpha=1.0, out_dtype=torch.bfloat16),
                         ops.attn_softmax_bwd(dr3, p3, s3, wl, ww, dwl, dww)))
    print(f'B {B} N {N} H {H} hd {hd}: mix fwd {t_f:7.1f} us (unfused nt+softmax {o_f:7.1f})   mix bwd {t_b:7.1f} us (unfused nt+softmax_bwd {o_b:7.1f})', flush=True)
