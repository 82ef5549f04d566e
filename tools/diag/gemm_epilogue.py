"""cost of the GEMM epilogue variants on the step's largest shapes (sustained, 300 launches each)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distillclip_amd import ops

def bench(fn, n=300):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3

for M, N, K in [(25600, 3072, 768), (25600, 768, 3072), (25600, 768, 768)]:
    a = torch.randn(M, K, device='cuda').bfloat16(); b = (torch.randn(N, K, device='cuda') * 0.03).bfloat16()
    bias = torch.randn(N, device='cuda')
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16); aux = torch.empty_like(out)
    z = torch.randn(M, N, device='cuda').bfloat16()
    outf = torch.empty(M, N, device='cuda'); res = torch.randn(M, N, device='cuda')
    cs = torch.zeros(N, device='cuda')
    cases = {
        'plain bf16': lambda: ops.gemm_nt(a, b, out=out),
        'bias': lambda: ops.gemm_nt(a, b, bias=bias, out=out),
        'bias+gelu': lambda: ops.gemm_nt(a, b, bias=bias, act='gelu', out=out),
        'bias+gelu+aux_out': lambda: ops.gemm_nt(a, b, bias=bias, act='gelu', aux_out=aux, out=out),
        'bias+quickgelu': lambda: ops.gemm_nt(a, b, bias=bias, act='quickgelu', out=out),
        'dgelu(aux_in)': lambda: ops.gemm_nt(a, b, act='dgelu', aux_in=z, out=out),
        'dgelu+colsum': lambda: ops.gemm_nt(a, b, act='dgelu', aux_in=z, out=out, colsum=cs),
        'f32 out': lambda: ops.gemm_nt(a, b, out=outf),
        'bias+residual f32': lambda: ops.gemm_nt(a, b, bias=bias, residual=res, out=outf),
        'residual in place f32': lambda: ops.gemm_nt(a, b, bias=bias, residual=outf, out=outf),
    }
    for name, fn in cases.items():
        t = bench(fn)
        print(f'{M:6d} {N:5d} {K:5d} {name:24s} {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s', flush=True)
