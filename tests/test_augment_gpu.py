"""HIP RandAugment + ToTensor + Normalize (dclip_augment_normalize through the C ABI) vs the real Pillow pipeline the
reference runs (oracle/augment.py pil_*; reference ms_coco.py:15-26, rand_augment.py).  Byte results are bit-exact; the
float32 output equals torch's ToTensor + Normalize bit for bit."""
import numpy as np
import pytest
import torch

from oracle import augment as A

pytestmark = pytest.mark.gpu


def _images(n, h, w, seed=0):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    out = []
    for i in range(n):
        kind = i % 4
        if kind == 0:
            img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        elif kind == 1:
            base = np.stack([(np.sin(x / (9.0 + i)) * 0.5 + 0.5) * 255, (np.cos(y / (13.0 + i)) * 0.5 + 0.5) * 200 + 20,
                             ((x + y + 7 * i) % 256)], -1)
            img = np.clip(base + rng.normal(0, 8, base.shape), 0, 255).astype(np.uint8)
        elif kind == 2:
            img = np.full((h, w, 3), (37 * i) % 256, dtype=np.uint8)
        else:
            img = rng.integers(90, 140, (h, w, 3), dtype=np.uint8)
        out.append(img)
    return np.stack(out)


def _check(imgs, plan, aug):
    got_f, got_b = aug(torch.from_numpy(imgs).cuda(), plan=plan, return_bytes=True)
    got_f, got_b = got_f.cpu().numpy(), got_b.cpu().numpy()
    for i, ops in enumerate(plan):
        want_b = A.pil_rand_augment(imgs[i], ops)
        assert np.array_equal(got_b[i], want_b), (i, ops, int((got_b[i] != want_b).sum()))
        assert np.array_equal(got_f[i], A.to_tensor_normalize(want_b)), (i, ops)


@pytest.mark.parametrize('h,w', [(224, 224), (336, 336), (96, 160)])
def test_every_op_both_signs_bit_exact(h, w):
    from distillclip_amd.augment import RandAugmentGPU
    plan = []
    for op in A.OPS:
        m = A.magnitude_of(op, 9, h, w)
        for sgn in ((1, -1) if op in A.SIGNED else (1,)):
            plan.append([(op, sgn * m)])
    imgs = _images(len(plan), h, w, seed=h)
    _check(imgs, plan, RandAugmentGPU(num_ops=1))
    # the same ops on a different image kind each (rotate the images against the plan)
    _check(np.roll(imgs, 1, axis=0), plan, RandAugmentGPU(num_ops=1))


def test_every_magnitude_bin_bit_exact():
    from distillclip_amd.augment import RandAugmentGPU
    plan = []
    for op in ('ShearX', 'ShearY', 'TranslateX', 'TranslateY', 'Rotate', 'Brightness', 'Contrast', 'Sharpness', 'Posterize'):
        for b in range(0, 31, 2):
            m = A.magnitude_of(op, b, 224, 224)
            plan.append([(op, -m if (b // 2) % 2 and op in A.SIGNED else m)])
    _check(_images(len(plan), 224, 224, seed=11), plan, RandAugmentGPU(num_ops=1))


def test_reference_chain_num_ops_4_bit_exact():
    """ms_coco.py:18: RandAugment(num_ops=4) with the draws torch's RNG yields for the reference loop"""
    from distillclip_amd.augment import RandAugmentGPU
    torch.manual_seed(2024)
    aug = RandAugmentGPU(num_ops=4)
    plan = aug.draw(64, 224, 224)
    assert len({name for ops in plan for name, _ in ops}) == 12                 # every op occurs in the sample
    _check(_images(64, 224, 224, seed=1), plan, aug)


def test_eval_transform_and_full_batch_properties():
    """B = 512 (the l_clip batch): Identity chain == eval transform == torch ToTensor + Normalize; posterize is idempotent;
    shifting there and back only blanks the border"""
    from distillclip_amd.augment import RandAugmentGPU, EvalTransformGPU
    B = 512
    imgs = _images(B, 224, 224, seed=9)
    x = torch.from_numpy(imgs).cuda()
    ev = EvalTransformGPU()(x)
    # the reference normalises in its DataLoader workers, i.e. with torch's CPU float32 kernels (IEEE divisions)
    xc = torch.from_numpy(imgs)
    want = (xc.permute(0, 3, 1, 2).float().div(255) - torch.tensor(A.IMAGE_MEAN).view(1, 3, 1, 1)) / \
        torch.tensor(A.IMAGE_STD).view(1, 3, 1, 1)
    assert torch.equal(ev.cpu(), want)
    aug = RandAugmentGPU(num_ops=2)
    ident, b0 = aug(x, plan=[[('Identity', 0.0), ('Identity', 0.0)]] * B, return_bytes=True)
    assert torch.equal(ident, ev) and torch.equal(b0, x)
    _, p1 = aug(x, plan=[[('Posterize', 5.0), ('Identity', 0.0)]] * B, return_bytes=True)
    _, p2 = aug(x, plan=[[('Posterize', 5.0), ('Posterize', 5.0)]] * B, return_bytes=True)
    assert torch.equal(p1, p2) and torch.equal(p1, x & 0xF8)
    _, sh = aug(x, plan=[[('TranslateX', 30.0), ('TranslateX', -30.0)]] * B, return_bytes=True)
    assert torch.equal(sh[:, :, :194], x[:, :, :194]) and int(sh[:, :, 194:].max()) == 0
    _, eq = aug(x, plan=[[('Equalize', 0.0), ('Equalize', 0.0)]] * B, return_bytes=True)
    assert eq.shape == x.shape


def test_argument_errors():
    from distillclip_amd.augment import normalize_batch, RandAugmentGPU
    with pytest.raises(RuntimeError):
        normalize_batch(torch.zeros(2, 8, 8, 3, dtype=torch.uint8))
    with pytest.raises(ValueError):
        normalize_batch(torch.zeros(2, 8, 8, 3, device='cuda'))
    with pytest.raises(ValueError):
        normalize_batch(torch.zeros(2, 2, 8, 3, dtype=torch.uint8, device='cuda'))          # H < 3
    with pytest.raises(ValueError):
        RandAugmentGPU(num_ops=1)(torch.zeros(2, 8, 8, 3, dtype=torch.uint8, device='cuda'), plan=[[('Invert', 0.0)]] * 2)


def test_device_prefetcher_order_and_values():
    """double-buffered H2D + GPU normalise: batches arrive in order, bit-equal to the direct transform"""
    from distillclip_amd.input_pipeline import DevicePrefetcher
    from distillclip_amd.augment import EvalTransformGPU, RandAugmentGPU
    rng = np.random.default_rng(4)
    batches = [(torch.from_numpy(rng.integers(0, 256, (8, 32, 48, 3), dtype=np.uint8)),
                torch.from_numpy(rng.integers(0, 400, (8, 77)).astype(np.int64))) for _ in range(5)]
    got = list(DevicePrefetcher(batches, train=False))
    assert len(got) == 5
    for (img, txt), (u8, ids) in zip(got, batches):
        assert torch.equal(txt.cpu(), ids)
        assert torch.equal(img, EvalTransformGPU()(u8.cuda()))
    # training mode: same generator seed -> same records as a direct call
    g1, g2 = torch.Generator().manual_seed(9), torch.Generator().manual_seed(9)
    tr = list(DevicePrefetcher(batches[:2], train=True, num_ops=4, generator=g1))
    aug = RandAugmentGPU(num_ops=4)
    for (img, _), (u8, _) in zip(tr, batches[:2]):
        assert torch.equal(img, aug(u8.cuda(), generator=g2))
    with pytest.raises(ValueError):
        list(DevicePrefetcher([(torch.zeros(2, 3, 8, 8), torch.zeros(2, 77, dtype=torch.int64))], train=False))


def test_resize_center_crop_mixed_sizes_bit_exact():
    """reference ms_coco.py:16-17 on a ragged batch (COCO-like sizes, tiny and large images, up-scaling)"""
    from PIL import Image
    from distillclip_amd.augment import ResizeCenterCropGPU
    sizes = [(480, 640), (640, 480), (427, 640), (375, 500), (500, 333), (224, 300), (300, 224), (224, 224), (100, 80),
             (80, 100), (1200, 1600), (225, 224), (2000, 230), (231, 229), (480, 640), (1, 5), (7, 3)]
    rng = np.random.default_rng(12)
    imgs = []
    for i, (h, w) in enumerate(sizes):
        if i % 2:
            y, x = np.mgrid[0:h, 0:w]
            a = np.stack([(x * 255 // max(w - 1, 1)), (y * 255 // max(h - 1, 1)), ((x + y) % 256)], -1).astype(np.uint8)
        else:
            a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        imgs.append(a)
    got = ResizeCenterCropGPU(224)(imgs).cpu().numpy()
    assert got.shape == (len(sizes), 224, 224, 3)
    for i, a in enumerate(imgs):
        want = np.asarray(A.pil_resize_center_crop(Image.fromarray(a, 'RGB'), 224))
        assert np.array_equal(got[i], want), (sizes[i], int((got[i] != want).sum()))
    # 336 px variant (BASELINE.json configs[3]) and torch tensors as input
    got336 = ResizeCenterCropGPU(336)([torch.from_numpy(imgs[0]), torch.from_numpy(imgs[10])]).cpu().numpy()
    for g, a in zip(got336, (imgs[0], imgs[10])):
        assert np.array_equal(g, np.asarray(A.pil_resize_center_crop(Image.fromarray(a, 'RGB'), 336)))


def test_full_train_chain_decode_to_tensor_bit_exact():
    """ms_coco.py:15-21 end to end on the GPU: Resize -> CenterCrop -> RandAugment(4) -> ToTensor -> Normalize"""
    from PIL import Image
    from distillclip_amd.augment import ResizeCenterCropGPU, RandAugmentGPU
    rng = np.random.default_rng(3)
    raw = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in [(480, 640), (333, 500), (640, 427), (256, 256)] * 4]
    torch.manual_seed(5)
    aug = RandAugmentGPU(num_ops=4)
    plan = aug.draw(len(raw), 224, 224)
    x = aug(ResizeCenterCropGPU(224)(raw), plan=plan).cpu().numpy()
    for i, a in enumerate(raw):
        img = np.asarray(A.pil_resize_center_crop(Image.fromarray(a, 'RGB'), 224))
        want = A.to_tensor_normalize(A.pil_rand_augment(img, plan[i]))
        assert np.array_equal(x[i], want), (i, plan[i])


def test_resize_argument_errors():
    from distillclip_amd.augment import ResizeCenterCropGPU
    with pytest.raises(ValueError):
        ResizeCenterCropGPU(224)([])
    with pytest.raises(ValueError):
        ResizeCenterCropGPU(224)([np.zeros((8, 8), dtype=np.uint8)])
    with pytest.raises(ValueError):
        ResizeCenterCropGPU(224)([np.zeros((8, 8, 3), dtype=np.float32)])
