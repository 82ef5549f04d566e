#!/usr/bin/env python3
"""dclip_attn_dqk (dQ and dK from one pass over dS) against dclip_attn_nn + dclip_attn_tn: equality of the outputs and time per call at the
step's student shapes (quad-blocked dS as the register-resident score stage writes it) and on row-major operands."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distillclip_amd._lib import lib


def run(B, H, N, hd, blocked, iters=50):
    D = H * hd
    Np = (N + 7) // 8 * 8
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device='cuda').manual_seed(N * 1000 + H)
    qkv = (torch.randn(B * N, 3 * D, device='cuda', generator=g) * 0.5).bfloat16()
    dS = torch.zeros(B, H, N, Np, device='cuda')
    dS[..., :N] = torch.randn(B, H, N, N, device='cuda', generator=g) * 0.1
    dS = dS.bfloat16()
    if blocked:      # [B, H, Np / 4, N, 4]
        dS_mem = dS.view(B, H, N, Np // 4, 4).permute(0, 1, 3, 2, 4).contiguous()
    else:
        dS_mem = dS.contiguous()
    q, k = qkv[:, :D], qkv[:, D:2 * D]
    out = [torch.zeros(B * N, 3 * D, device='cuda', dtype=torch.bfloat16) for _ in range(2)]
    scale = hd ** -0.5

    def two():
        lib().dclip_attn_nn(dS_mem.data_ptr(), k.data_ptr(), 3 * D, out[0].data_ptr(), 3 * D, B, H, N, Np, hd, scale, int(blocked), st)
        lib().dclip_attn_tn(dS_mem.data_ptr(), q.data_ptr(), 3 * D, out[0][:, D:].data_ptr(), 3 * D, B, H, N, Np, hd, scale, int(blocked), st)

    def one():
        lib().dclip_attn_dqk(dS_mem.data_ptr(), q.data_ptr(), k.data_ptr(), 3 * D, out[1].data_ptr(), out[1][:, D:].data_ptr(), 3 * D, B, H, N, Np, hd,
                             scale, int(blocked), st)
    two(); one()
    torch.cuda.synchronize()
    same = torch.equal(out[0], out[1])
    ref_q = torch.einsum('bhij,bjhd->bihd', dS[..., :N].float(), k.float().view(B, N, H, hd)).reshape(B * N, D) * scale
    err = ((out[1][:, :D].float() - ref_q).norm() / ref_q.norm()).item()
    res = {}
    for name, fn in (('nn+tn', two), ('dqk', one)):
        for _ in range(5):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        res[name] = a.elapsed_time(b) / iters * 1e3
    print(f'B={B} H={H} N={N} hd={hd} blocked={blocked}: identical={same} dQ rel err vs f32 {err:.2e}  nn+tn {res["nn+tn"]:.1f} us  dqk {res["dqk"]:.1f} us', flush=True)
    return same


if __name__ == '__main__':
    ok = True
    for shape in [(512, 24, 50, 32, True), (512, 12, 77, 64, True), (512, 24, 101, 32, True), (512, 12, 50, 64, False), (512, 8, 77, 64, False),
                  (3, 2, 13, 64, False), (5, 3, 17, 32, True), (2, 4, 128, 32, False), (2, 1, 128, 64, True), (4, 8, 65, 64, True)]:
        ok = run(*shape) and ok
    sys.exit(0 if ok else 1)
