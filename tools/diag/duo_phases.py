"""in-kernel stamps of gemm_nt_duo_kernel (dclip_trace_gemm_stamps; 8 u64 per workgroup): prologue / main loop / epilogue cycles per
workgroup, split by the priority the workgroup took, the CU occupancy pattern (HW_ID / XCC_ID census) and how much of the epilogues ran
beside another workgroup's main loop on the same CU.  DCLIP_GEMM_DUO=2 python tools/diag/duo_phases.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distillclip_amd import ops
from distillclip_amd._lib import lib
shapes = [(25600, 2304, 768, 'bf16'), (25600, 768, 768, 'res'), (25600, 3072, 768, 'qgelu'), (25600, 768, 3072, 'res'), (39424, 512, 512, 'res'),
          (39424, 3072, 768, 'gelu_save')]
for M, N, K, kind in shapes:
    a = torch.randn(M, K, device='cuda').bfloat16(); b = (torch.randn(N, K, device='cuda') * 0.05).bfloat16()
    bias = torch.randn(N, device='cuda'); res = torch.randn(M, N, device='cuda')
    aux = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    def run():
        if kind == 'res': ops.gemm_nt(a, b, bias=bias, residual=res, out=res)
        elif kind == 'qgelu': ops.gemm_nt(a, b, bias=bias, act='quickgelu')
        elif kind == 'gelu_save': ops.gemm_nt(a, b, bias=bias, act='gelu_save', aux_out=aux)
        else: ops.gemm_nt(a, b)
    for _ in range(20): run()
    buf = torch.zeros(8 * 4096, dtype=torch.int64, device='cuda')
    lib().dclip_trace_gemm_stamps(buf.data_ptr())
    run(); torch.cuda.synchronize()
    lib().dclip_trace_gemm_stamps(None)
    s = buf.view(-1, 8).cpu()
    s = s[s[:, 0] != 0]
    n = s.shape[0]
    d = s.double()
    pro, main, epi = (d[:, 1] - d[:, 0]), (d[:, 2] - d[:, 1]), (d[:, 3] - d[:, 2])
    clk = ((d[:, 3] - d[:, 0]) / (d[:, 5] - d[:, 4]).clamp(min=1) * 0.1).median().item()
    span = (d[:, 5].max() - d[:, 4].min()).item() * 0.01
    hw = s[:, 6] & 0xffffffff; hi = (s[:, 6] >> 32) & 1; xcc = s[:, 7] & 0xf
    wave_slot = hw & 0xf; simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    cuid = (xcc * 8 + se) * 32 + sh * 16 + cu
    per_cu = collections.Counter(cuid.tolist())
    nk = K // 32
    print(f'{M:6d} {N:5d} {K:5d} {kind:9s} wgs {n:5d} on {len(per_cu)} CUs (max {max(per_cu.values())} per CU) clock {clk:4.2f} GHz span {span:7.1f} us', flush=True)
    for h in (1, 0):
        m = hi == h
        if m.any():
            print(f'    prio {"hi" if h else "lo"}: {int(m.sum()):5d} wgs | median cycles: prologue {pro[m].median().item():7.0f}  main {main[m].median().item():8.0f} '
                  f'({main[m].median().item() / nk:6.0f} / stage)  epilogue {epi[m].median().item():7.0f} | wave slots {sorted(collections.Counter(wave_slot[m].tolist()).items())}')
    # overlap: for every workgroup, the fraction of its epilogue interval [t2, t3] (real time from cycle stamps is per-CU consistent: s_memtime
    # is a per-XCD counter) during which another workgroup of the SAME CU was in its main loop [t1, t2]
    by_cu = collections.defaultdict(list)
    for i in range(n):
        by_cu[int(cuid[i])].append((d[i, 1].item(), d[i, 2].item(), d[i, 3].item()))
    cov, tote = 0.0, 0.0
    for wl in by_cu.values():
        for i, (t1, t2, t3) in enumerate(wl):
            tote += t3 - t2
            for j, (u1, u2, u3) in enumerate(wl):
                if i != j:
                    cov += max(0.0, min(t3, u2) - max(t2, u1))
    print(f'    epilogue time covered by a neighbour\'s main loop on the same CU: {cov / max(tote, 1):.2f}')
