"""validation retrieval metrics: HIP kernel time at the COCO-val size vs the oracle on the host cores."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distillclip_amd.metrics import retrieval_metrics
from oracle import metrics as om

for n, E in [(5000, 512), (40000, 512)]:
    g = torch.Generator().manual_seed(1)
    img = torch.randn(n, E, generator=g); txt = 0.1 * img + torch.randn(n, E, generator=g)
    a, b = img.cuda(), txt.cuda()
    for _ in range(2): retrieval_metrics(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10 if n <= 5000 else 2
    e0.record()
    for _ in range(reps): m = retrieval_metrics(a, b)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    line = f'n={n} E={E}: HIP {t*1e3:.3f} ms  ({2*n*n*E/t/1e12:.1f} TFLOP/s exact-f32 MFMA, logits never stored)'
    if n <= 5000:
        t0 = time.time(); w = om.retrieval_metrics(img, txt); tc = time.time() - t0
        line += f' | oracle (torch f64, {torch.get_num_threads()} threads) {tc*1e3:.0f} ms | acc@1 {m["acc_top1"].item():.4f} vs {w["acc_top1"].item():.4f}'
    print(line, flush=True)
