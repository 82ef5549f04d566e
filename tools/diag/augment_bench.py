"""GPU RandAugment + normalise at the l_clip batch vs the reference's PIL chain on the host cores."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from distillclip_amd.augment import RandAugmentGPU
from oracle import augment as A

B = 512
rng = np.random.default_rng(0)
imgs = rng.integers(0, 256, (B, 224, 224, 3), dtype=np.uint8)
x = torch.from_numpy(imgs).cuda()
torch.manual_seed(0)
aug = RandAugmentGPU(num_ops=4)
plan = aug.draw(B, 224, 224)
rec = aug.records(plan, 224, 224)
from distillclip_amd.augment import normalize_batch
for _ in range(3): normalize_batch(x, rec)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): normalize_batch(x, rec)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20 * 1e-3
nb = B * 224 * 224 * 3
print(f'HIP: {t*1e3:.3f} ms / batch of {B}  = {B/t:,.0f} images/s ; bytes in+out {nb*5/t/1e12:.2f} TB/s algorithmic (u8 in, f32 out)')
t0 = time.time()
for i in range(64):
    A.to_tensor_normalize(A.pil_rand_augment(imgs[i], plan[i]))
tc = (time.time() - t0) / 64
print(f'PIL chain (1 core): {tc*1e3:.2f} ms / image = {1/tc:,.0f} images/s per core')
t0 = time.time(); aug.records(aug.draw(B, 224, 224), 224, 224); print(f'host draw + records for {B} (reference RNG order): {(time.time()-t0)*1e3:.1f} ms')
aug.draw_records(B, 224, 224)
t0 = time.time(); aug.draw_records(B, 224, 224); print(f'host bulk draw_records for {B}: {(time.time()-t0)*1e3:.3f} ms')

# resize + crop of a COCO-like ragged batch
from PIL import Image
from distillclip_amd.augment import ResizeCenterCropGPU
raw = [rng.integers(0, 256, s + (3,), dtype=np.uint8) for s in [(480, 640), (640, 480), (427, 640), (500, 375)] * (B // 4)]
rc = ResizeCenterCropGPU(224)
rc(raw); torch.cuda.synchronize()
t0 = time.time(); y = rc(raw); torch.cuda.synchronize(); tg = time.time() - t0
t0 = time.time()
for a in raw[:32]: A.pil_resize_center_crop(Image.fromarray(a, 'RGB'), 224)
tp = (time.time() - t0) / 32
print(f'resize+crop: GPU path (pack + H2D + kernel, {sum(a.size for a in raw)/1e6:.0f} MB of pixels) {tg*1e3:.1f} ms / {B} images = {B/tg:,.0f} images/s ; PIL {tp*1e3:.2f} ms / image per core')
