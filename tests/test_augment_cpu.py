"""RandAugment restatement (oracle/augment.py np_*: the pixel rules the HIP kernel implements) against the real Pillow calls
behind the reference's torchvision chain (oracle/augment.py pil_*; reference rand_augment.py:10-87, ms_coco.py:15-26), and the
host-side op records (distillclip_amd/augment.py)."""
import numpy as np
import pytest
import torch

from oracle import augment as A


def _images(h, w, seed=0):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([(np.sin(x / 17.0) * 0.5 + 0.5) * 255, (np.cos(y / 23.0) * 0.5 + 0.5) * 200 + 20, ((x + y) % 256)], -1)
    return {'noise': rng.integers(0, 256, (h, w, 3), dtype=np.uint8),
            'smooth': np.clip(base + rng.normal(0, 8, base.shape), 0, 255).astype(np.uint8),
            'flat': np.full((h, w, 3), 77, dtype=np.uint8),
            'lowcontrast': rng.integers(90, 140, (h, w, 3), dtype=np.uint8)}


@pytest.mark.parametrize('op', A.OPS)
def test_numpy_rules_equal_pillow_every_bin(op):
    for name, img in _images(224, 224).items():
        for b in range(31):
            m = A.magnitude_of(op, b, 224, 224)
            for sgn in ((1, -1) if op in A.SIGNED else (1,)):
                want = A.pil_rand_augment(img, [(op, sgn * m)])
                got = A.np_apply_op(img, op, sgn * m)
                assert np.array_equal(want, got), (name, op, b, sgn, int((want != got).sum()))
        if op in ('Identity', 'AutoContrast', 'Equalize'):
            break                                    # magnitude-free ops: one pass over the images is the whole space


@pytest.mark.parametrize('h,w', [(336, 336), (96, 160), (33, 21)])
def test_numpy_rules_equal_pillow_other_sizes(h, w):
    for name, img in _images(h, w, seed=3).items():
        for op in A.OPS:
            m = A.magnitude_of(op, 9, h, w)
            for sgn in ((1, -1) if op in A.SIGNED else (1,)):
                want = A.pil_rand_augment(img, [(op, sgn * m)])
                assert np.array_equal(want, A.np_apply_op(img, op, sgn * m)), (name, op, sgn)


def test_chains_of_four_ops_equal_pillow():
    from distillclip_amd.augment import RandAugmentGPU
    torch.manual_seed(7)
    aug = RandAugmentGPU(num_ops=4)
    plan = aug.draw(24, 224, 224)
    imgs = list(_images(224, 224, seed=5).values())
    for i, ops in enumerate(plan):
        img = imgs[i % len(imgs)]
        assert np.array_equal(A.pil_rand_augment(img, ops), A.np_rand_augment(img, ops)), ops


def test_draw_consumes_rng_like_the_reference_loop():
    """reference rand_augment.py:152-164: per image, per op: randint(12), then randint(2) only for signed ops"""
    from distillclip_amd.augment import RandAugmentGPU, OP_NAMES, augmentation_space
    assert OP_NAMES == A.OPS
    torch.manual_seed(123)
    plan = RandAugmentGPU(num_ops=4).draw(5, 224, 224)
    torch.manual_seed(123)
    meta = augmentation_space(31, 224, 224)
    for ops in plan:
        for name, mag in ops:
            idx = int(torch.randint(len(meta), (1,)).item())
            assert OP_NAMES[idx] == name
            table, signed = meta[name]
            m = float(table[9].item()) if table.ndim > 0 else 0.0
            if signed and torch.randint(2, (1,)):
                m *= -1.0
            assert m == mag
            assert abs(abs(mag) - A.magnitude_of(name, 9, 224, 224)) < 1e-12


def test_op_records_match_the_oracle_coefficients():
    from distillclip_amd.augment import op_record
    for op in ('ShearX', 'ShearY', 'Rotate'):
        for sgn in (1, -1):
            m = sgn * A.magnitude_of(op, 9, 224, 224)
            rec = op_record(op, m, 224, 224)
            assert int(rec['op']) == 1
            assert tuple(int(v) for v in rec['c']) == A.affine_fixed_coeffs(A.op_matrix(op, m, 224, 224))
    rec = op_record('TranslateX', -30.4, 224, 224)
    assert int(rec['op']) == 2 and (int(rec['c'][0]), int(rec['c'][1])) == (30, 0)          # source = x - tx, tx = int(-30.4)
    rec = op_record('TranslateY', 30.4, 224, 224)
    assert int(rec['op']) == 2 and (int(rec['c'][0]), int(rec['c'][1])) == (0, -30)
    assert int(op_record('ShearX', 0.0, 224, 224)['op']) == 2                               # identity matrix: zero shift
    assert int(op_record('Posterize', 7.0, 224, 224)['c'][0]) == 0xFE
    assert abs(float(op_record('Contrast', -0.27, 224, 224)['f']) - 0.73) < 1e-7
    with pytest.raises(ValueError):
        op_record('Solarize', 1.0, 224, 224)


def test_to_tensor_normalize_matches_plain_torch():
    img = _images(224, 224)['noise']
    got = A.to_tensor_normalize(img)
    want = (torch.from_numpy(img).permute(2, 0, 1).float() / 255 - torch.tensor(A.IMAGE_MEAN).view(3, 1, 1)) / torch.tensor(A.IMAGE_STD).view(3, 1, 1)
    assert np.array_equal(got, want.numpy())


def test_bulk_draw_has_the_reference_distribution_and_records():
    from distillclip_amd.augment import RandAugmentGPU, op_record, OP_NAMES, AUG_OP_DTYPE
    aug = RandAugmentGPU(num_ops=4)
    g = torch.Generator().manual_seed(3)
    rec = aug.draw_records(4096, 224, 224, generator=g)
    assert rec.shape == (4096, 4) and rec.dtype == AUG_OP_DTYPE
    # every record is one of the 20 records the per-op path can produce at bin 9
    legal = set()
    for name in OP_NAMES:
        m = A.magnitude_of(name, 9, 224, 224)
        for sgn in ((1, -1) if name in A.SIGNED else (1,)):
            legal.add(op_record(name, sgn * m, 224, 224).tobytes())
    seen = {r.tobytes() for r in rec.reshape(-1)}
    assert seen <= legal and len(seen) == len(legal)
    # op frequencies: uniform over the 12 ops (identity / zero-magnitude records collapse by op code)
    codes = rec['op'].reshape(-1)
    frac_identity = float((codes == 0).mean())
    assert abs(frac_identity - 1 / 12) < 0.01
    assert abs(float((codes == 7).mean()) - 1 / 12) < 0.01 and abs(float((codes == 3).mean()) - 1 / 12) < 0.01


SIZES = [(480, 640), (640, 480), (427, 640), (375, 500), (500, 333), (224, 300), (300, 224), (224, 224), (100, 80),
         (80, 100), (1200, 1600), (225, 224), (2000, 230), (231, 229)]


@pytest.mark.parametrize('h,w', SIZES)
def test_resize_center_crop_restatement_equals_pillow(h, w):
    """reference ms_coco.py:16-17: Resize(224) + CenterCrop(224) on PIL images (down- and up-scaling, both orientations)"""
    from PIL import Image
    arr = np.random.default_rng(h * 7 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    want = np.asarray(A.pil_resize_center_crop(Image.fromarray(arr, 'RGB'), 224))
    assert np.array_equal(want, A.np_resize_center_crop(arr, 224))


@pytest.mark.parametrize('h,w', SIZES + [(3000, 4000)])
def test_host_resample_tables_equal_the_oracle_coefficients(h, w):
    """the product's vectorised table builder (distillclip_amd/augment.py) against the scalar restatement of Resample.c"""
    from distillclip_amd.augment import resample_tables
    block, row0, nrows, ksh, ksv = resample_tables(h, w, 224)
    nw, nh = A.resize_target(w, h, 224)
    left, top = int(round((nw - 224) / 2.0)), int(round((nh - 224) / 2.0))
    hb, hk = A.resample_coeffs(w, nw)
    vb, vk = A.resample_coeffs(h, nh)
    hb, hk, vb, vk = hb[left:left + 224], hk[left:left + 224], vb[top:top + 224].copy(), vk[top:top + 224]
    assert row0 == int(vb[0, 0]) and nrows == int(vb[-1, 0] + vb[-1, 1]) - row0
    vb[:, 0] -= row0
    want = np.concatenate([hb.reshape(-1), hk.reshape(-1), vb.reshape(-1), vk.reshape(-1)])
    assert (ksh, ksv) == (hk.shape[1], vk.shape[1])
    assert np.array_equal(block, want)
