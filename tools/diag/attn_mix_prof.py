"""dclip_attn_mix_fwd / _bwd alone on the step's two student shapes (preallocated outputs, no torch work in the timed loop);
run it under rocprofv3 (--kernel-trace --stats, or --pmc ...) to read kernel time / counters.  argv: [reps] [which: all|img|txt]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distillclip_amd._lib import lib
from distillclip_amd.ops import _p, _stream

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
which = sys.argv[2] if len(sys.argv) > 2 else 'all'
if which == 'teacher':
    from distillclip_amd import ops
    for B, N, H, hd, causal in [(512, 50, 12, 64, False), (512, 77, 8, 64, True)]:
        D = H * hd
        qkv = (torch.randn(B * N, 3 * D) * 0.7).bfloat16().cuda()
        fn = lambda: ops.attn_fused_fwd(qkv, B, N, H, hd, causal)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        print(f'teacher fused attention B {B} N {N} H {H} hd {hd} causal {causal}: {a.elapsed_time(b) / reps * 1e3:8.1f} us (incl. output alloc)', flush=True)
    sys.exit(0)
shapes = [(512, 50, 24, 32)] * (which in ('all', 'img')) + [(512, 77, 12, 64)] * (which in ('all', 'txt'))
for B, N, H, hd in shapes:
    D = H * hd
    Np = (N + 7) // 8 * 8
    g = torch.Generator(device='cpu').manual_seed(1)
    qkv = (torch.randn(B * N, 3 * D, generator=g) * 0.7).bfloat16().cuda()
    dctx = torch.randn(B * N, D, generator=g).bfloat16().cuda()
    wl = (torch.eye(H) + 0.1 * torch.randn(H, H, generator=g)).cuda()
    ww = (torch.eye(H) + 0.1 * torch.randn(H, H, generator=g)).cuda()
    scale = hd ** -0.5
    R = torch.empty(B, H, Np // 4, N, 4, dtype=torch.bfloat16, device='cuda')
    dS = torch.empty_like(R)
    lse = torch.empty(B, H, N, device='cuda')
    dwl, dww = torch.zeros(H, H, device='cuda'), torch.zeros(H, H, device='cuda')
    ws = torch.empty(lib().dclip_attn_mix_bwd_workspace_bytes(B, H, N), dtype=torch.uint8, device='cuda')
    fwd = lambda: lib().dclip_attn_mix_fwd(_p(qkv), 3 * D, _p(wl), _p(ww), _p(R), _p(lse), B, H, N, Np, hd, scale, _stream())
    bwd = lambda: lib().dclip_attn_mix_bwd(_p(qkv), 3 * D, _p(dctx), D, _p(wl), _p(ww), _p(lse), _p(dS), _p(dwl), _p(dww), _p(ws), ws.numel(),
                                           B, H, N, Np, hd, scale, _stream())
    if os.environ.get('MIX_STAMPS') == '1':
        QT = (N + 15) // 16
        st = torch.zeros(B * QT, 12, dtype=torch.int64, device='cuda')
        fwd(); torch.cuda.synchronize()
        sa, sb = torch.zeros(B * QT, 8, dtype=torch.int64, device='cuda'), torch.zeros(B * QT, 8, dtype=torch.int64, device='cuda')
        bwd(); torch.cuda.synchronize()
        lib().dclip_attn_mix_debug_stamps(_p(st), _p(sa), _p(sb))
        fwd(); bwd(); torch.cuda.synchronize()
        lib().dclip_attn_mix_debug_stamps(None, None, None)
        for nm, t in (('bwd pass A', sa), ('bwd pass B', sb)):
            print('  %s stamps (median cycles per tile): total %d ring wait %d S+dR %d per-key stage %d dW product %d' %
                  ((nm,) + tuple(t.double().median(0).values.tolist()[:5])), flush=True)
        med = st.double().median(0).values.tolist()
        print('  fwd stamps (median cycles per tile): pass1 total %d wait %d scores %d stage %d | pass2 total %d wait %d scores %d stage %d' % tuple(med[:8]), flush=True)
        rt = st[:, 10:12].double()
        print('  prologue %d cycles, query fragments %d cycles; wave lifetime median %.1f us, first start -> last end %.1f us, start spread %.1f us'
              % (med[8], med[9], (rt[:, 1] - rt[:, 0]).median().item() / 100, (rt[:, 1].max() - rt[:, 0].min()).item() / 100,
                 (rt[:, 0].max() - rt[:, 0].min()).item() / 100), flush=True)
    for name, fn in (('fwd', fwd), ('bwd', bwd)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        print(f'B {B} N {N} H {H} hd {hd} mix {name}: {a.elapsed_time(b) / reps * 1e3:8.1f} us', flush=True)
