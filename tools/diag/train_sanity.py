"""40 training steps of the l_clip dual step at the bench shapes (B=128): loss must decrease, parameters stay finite."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from distillclip_amd import synth
dev = torch.device('cuda', 0)
model = bench.build_model(2022, dev)
(opt,), (sched,) = model.configure_optimizers()
opt.lr = opt.base_lr = 3e-4
B = 128
image = torch.from_numpy(synth.images(1, B)).to(dev); text = torch.from_numpy(synth.captions(1, B)).to(dev)
losses = []
for it in range(40):
    loss = model.training_step([image, text])
    opt.zero_grad()
    model.backward_and_sync(loss)
    opt.step()
    if it % 5 == 0 or it == 39:
        losses.append(round(loss.item(), 5))
        print(it, losses[-1], {k: round(v.item(), 5) for k, v in model.last_cal_res.items()}, flush=True)
ok = all(torch.isfinite(p).all().item() for p in model.student.parameters())
print('finite', ok, 'decreased', losses[-1] < losses[0] * 0.9, 'mem GB', torch.cuda.max_memory_allocated() / 2**30)
assert ok and losses[-1] < losses[0] * 0.9
