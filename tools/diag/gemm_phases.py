"""where a 256-row / 320-row gemm_nt workgroup spends its cycles (in-kernel s_memtime stamps, diagnostic launches):
prologue (first operands landed) / main loop / epilogue, per workgroup, plus the clock the chip held."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distillclip_amd import ops
from distillclip_amd._lib import lib
shapes = [(2048, 1024, 768, 'bf16'), (2048, 1024, 768, 'qgelu'), (2048, 1024, 768, 'res'), (8192, 2048, 768, 'bf16'), (8192, 2048, 768, 'res'), (25600, 3072, 768, 'qgelu'), (25600, 2304, 768, 'bf16'), (25600, 768, 3072, 'res'), (25600, 768, 768, 'res'), (39424, 3072, 768, 'gelu_save'),
          (39424, 768, 3072, 'bf16'), (39424, 512, 2048, 'res')]
for M, N, K, kind in shapes:
    a = torch.randn(M, K, device='cuda').bfloat16(); b = (torch.randn(N, K, device='cuda') * 0.05).bfloat16()
    bias = torch.randn(N, device='cuda'); res = torch.randn(M, N, device='cuda'); out = torch.empty(M, N, device='cuda')
    aux = torch.empty(M, N, device='cuda', dtype=torch.uint8)
    def run():
        if kind == 'res': ops.gemm_nt(a, b, bias=bias, residual=res, out=out)
        elif kind == 'qgelu': ops.gemm_nt(a, b, bias=bias, act='quickgelu')
        elif kind == 'gelu_save': ops.gemm_nt(a, b, bias=bias, act='gelu_save', aux_out=aux)
        else: ops.gemm_nt(a, b)
    for _ in range(20): run()
    buf = torch.zeros(6 * 4096, dtype=torch.int64, device='cuda')
    lib().dclip_trace_gemm_stamps(buf.data_ptr())
    run(); torch.cuda.synchronize()
    lib().dclip_trace_gemm_stamps(None)
    s = buf.view(-1, 6).cpu()
    s = s[s[:, 0] != 0].double()
    n = s.shape[0]
    pro, main, epi = (s[:, 1] - s[:, 0]), (s[:, 2] - s[:, 1]), (s[:, 3] - s[:, 2])
    clk = ((s[:, 3] - s[:, 0]) / (s[:, 5] - s[:, 4]).clamp(min=1) * 0.1).median().item()      # GHz: memtime cycles per 10 ns tick
    span = (s[:, 5].max() - s[:, 4].min()).item() * 0.01
    nk = K // 64
    print(f'{M:6d} {N:5d} {K:5d} {kind:9s} wgs {n:5d} clock {clk:4.2f} GHz  kernel span {span:7.1f} us | per workgroup (median cycles): '
          f'prologue {pro.median().item():7.0f}  main {main.median().item():8.0f} ({main.median().item() / nk:6.0f} / k-tile)  epilogue {epi.median().item():7.0f} '
          f'| sum {(pro + main + epi).median().item() / clk / 1e3:6.1f} us', flush=True)
