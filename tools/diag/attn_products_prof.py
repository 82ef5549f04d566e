"""dclip_attn_nn / _tn (quad-blocked score operand, as the default student path issues them) and the teacher's fused attention on
the step's shapes, HIP-event timed.  python tools/diag/attn_products_prof.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distillclip_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for B, N, H, hd in [(512, 50, 24, 32), (512, 77, 12, 64)]:
    D = H * hd
    Np = (N + 7) // 8 * 8
    R = torch.randn(B, H, Np // 4, N, 4, device='cuda').bfloat16()
    x = torch.randn(B * N, 3 * D, device='cuda').bfloat16()
    out = torch.empty(B * N, D, dtype=torch.bfloat16, device='cuda')
    mb = (2 * B * H * N * Np + 4 * B * N * D) / 1e6
    for name, fn in (('nn', lambda: ops.attn_nn(R, x[:, 2 * D:], 3 * D, out, D, hd)),
                     ('tn', lambda: ops.attn_tn(R, x[:, :D], 3 * D, out, D, hd))):
        us = timed(fn)
        print(f'attn_{name} blocked B {B} N {N} H {H} hd {hd}: {us:7.1f} us  {mb / us:6.2f} TB/s algorithmic', flush=True)
for B, N, H, hd, causal in [(512, 50, 12, 64, False), (512, 77, 8, 64, True)]:
    D = H * hd
    qkv = (torch.randn(B * N, 3 * D, device='cuda') * 0.7).bfloat16()
    us = timed(lambda: ops.attn_fused_fwd(qkv, B, N, H, hd, causal))
    print(f'teacher fused B {B} N {N} H {H} hd {hd} causal {causal}: {us:7.1f} us  {8 * B * N * D / 1e6 / us:6.2f} TB/s algorithmic (incl. output alloc)', flush=True)
