"""sustained-load GEMM throughput: is the in-step GEMM rate set by clocks / power rather than by the kernel schedule?"""
import sys, os, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distillclip_amd import ops

M, N, K = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (25600, 3072, 768))]
a = torch.randn(M, K, device='cuda').bfloat16(); b = torch.randn(N, K, device='cuda').bfloat16()
out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
for _ in range(3): ops.gemm_nt(a, b, out=out)
torch.cuda.synchronize()
chunk = 500
for rep in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(chunk): ops.gemm_nt(a, b, out=out)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / chunk * 1e-3
    smi = ''
    if rep % 3 == 2:
        try:
            smi = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=20).stdout
            smi = ' | '.join(l.strip() for l in smi.splitlines() if ('sclk' in l or 'Power' in l or 'mclk' in l))[:300]
        except Exception as ex:
            smi = repr(ex)
    print(f'chunk {rep:2d}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s  {smi}', flush=True)
