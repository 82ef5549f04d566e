"""GPU image transform of the training set: the part of the reference's per-sample torchvision chain
that follows decode / resize / crop — RandAugment(num_ops=4) -> ToTensor -> Normalize (reference
data/component/ms_coco.py:15-26, rand_augment.py, utils.py:11-12) — as one HIP kernel over a uint8 batch
(`dclip_augment_normalize`), bit-identical to the PIL pipeline for the same op draws.

    aug = RandAugmentGPU(num_ops=4)                      # same constructor arguments as the reference's RandAugment
    x = aug(batch_u8)                                    # uint8 [B,H,W,3] on the GPU -> float32 [B,3,H,W], normalised

The op draws consume torch's global CPU generator exactly like the reference's loop (one randint(12) per op, one randint(2)
per signed op), image by image.
"""
import math

import numpy as np
import torch

from ._lib import lib

IMAGE_MEAN = (0.48145466, 0.4578275, 0.40821073)         # reference data/component/utils.py:11-12
IMAGE_STD = (0.26862954, 0.26130258, 0.27577711)

OP_NAMES = ['Identity', 'ShearX', 'ShearY', 'TranslateX', 'TranslateY', 'Rotate', 'Brightness', 'Contrast', 'Sharpness',
            'Posterize', 'AutoContrast', 'Equalize']     # key order of rand_augment.py:128-143
_SIGNED = {'ShearX', 'ShearY', 'TranslateX', 'TranslateY', 'Rotate', 'Brightness', 'Contrast', 'Sharpness'}
_CODE = {'Identity': 0, 'affine': 1, 'shift': 2, 'Brightness': 3, 'Contrast': 4, 'Sharpness': 5, 'Posterize': 6,
         'AutoContrast': 7, 'Equalize': 8}
AUG_OP_DTYPE = np.dtype([('op', '<i4'), ('c', '<i4', (6,)), ('f', '<f4')])      # struct dclip_aug_op (include/dclip.h)


def augmentation_space(num_bins, height, width):
    """rand_augment.py:128-143: op -> (magnitude table, signed)"""
    return {
        'Identity': (torch.tensor(0.0), False),
        'ShearX': (torch.linspace(0.0, 0.3, num_bins), True),
        'ShearY': (torch.linspace(0.0, 0.3, num_bins), True),
        'TranslateX': (torch.linspace(0.0, 150.0 / 331.0 * width, num_bins), True),
        'TranslateY': (torch.linspace(0.0, 150.0 / 331.0 * height, num_bins), True),
        'Rotate': (torch.linspace(0.0, 30.0, num_bins), True),
        'Brightness': (torch.linspace(0.0, 0.9, num_bins), True),
        'Contrast': (torch.linspace(0.0, 0.9, num_bins), True),
        'Sharpness': (torch.linspace(0.0, 0.9, num_bins), True),
        'Posterize': (8 - (torch.arange(num_bins) / ((num_bins - 1) / 4)).round().int(), False),
        'AutoContrast': (torch.tensor(0.0), False),
        'Equalize': (torch.tensor(0.0), False),
    }


def _inverse_affine(center, angle, translate, scale, shear):
    # torchvision.transforms.functional._get_inverse_affine_matrix
    rot, sx, sy = math.radians(angle), math.radians(shear[0]), math.radians(shear[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d / scale, -b / scale, 0.0, -c / scale, a / scale, 0.0]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


def _rotate_matrix(angle, w, h):
    # PIL.Image.Image.rotate (expand=False, centre = image centre)
    cx, cy = w / 2, h / 2
    ang = -math.radians(angle % 360.0)
    m = [round(math.cos(ang), 15), round(math.sin(ang), 15), 0.0, round(-math.sin(ang), 15), round(math.cos(ang), 15), 0.0]
    m[2] = m[0] * -cx + m[1] * -cy + m[2] + cx
    m[5] = m[3] * -cx + m[4] * -cy + m[5] + cy
    return m


def _geometric_record(m, rec):
    """Image.transform(AFFINE, NEAREST): Pillow scales/translates (a1 == a3 == 0) on its ImagingScaleAffine path, everything
    else on the 16.16 fixed-point path."""
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
    if m[1] == 0 and m[3] == 0:
        xo, yo = m[2] + m[0] * 0.5, m[5] + m[4] * 0.5
        if m[0] != 1.0 or m[4] != 1.0 or (xo - 0.5) != int(xo - 0.5) or (yo - 0.5) != int(yo - 0.5):
            raise NotImplementedError('axis-aligned affine with scale != 1 or a fractional offset (not produced by RandAugment)')
        rec['op'] = _CODE['shift']
        rec['c'][0], rec['c'][1] = int(xo - 0.5), int(yo - 0.5)
        return
    rec['op'] = _CODE['affine']
    rec['c'][:] = [fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]),
                   fix(m[5] + m[3] * 0.5 + m[4] * 0.5)]


def op_record(name, magnitude, height, width):
    """one `dclip_aug_op` for rand_augment.py:10-87 `_apply_op(img, name, magnitude)` on a height x width image"""
    rec = np.zeros((), dtype=AUG_OP_DTYPE)
    w, h = width, height
    if name == 'Identity':
        pass
    elif name == 'ShearX':
        _geometric_record(_inverse_affine([0, 0], 0.0, [0, 0], 1.0, [math.degrees(math.atan(magnitude)), 0.0]), rec)
    elif name == 'ShearY':
        _geometric_record(_inverse_affine([0, 0], 0.0, [0, 0], 1.0, [0.0, math.degrees(math.atan(magnitude))]), rec)
    elif name == 'TranslateX':
        _geometric_record(_inverse_affine([w * 0.5, h * 0.5], 0.0, [int(magnitude), 0], 1.0, [0.0, 0.0]), rec)
    elif name == 'TranslateY':
        _geometric_record(_inverse_affine([w * 0.5, h * 0.5], 0.0, [0, int(magnitude)], 1.0, [0.0, 0.0]), rec)
    elif name == 'Rotate':
        ang = magnitude % 360.0
        if ang in (0.0, 90.0, 180.0, 270.0):
            if ang != 0.0:
                raise NotImplementedError('quarter-turn rotations take PIL.transpose (not produced by RandAugment magnitudes)')
        else:
            _geometric_record(_rotate_matrix(magnitude, w, h), rec)
    elif name in ('Brightness', 'Contrast', 'Sharpness'):
        rec['op'] = _CODE[name]
        rec['f'] = 1.0 + magnitude
    elif name == 'Posterize':
        rec['op'] = _CODE[name]
        rec['c'][0] = ~(2 ** (8 - int(magnitude)) - 1) & 0xFF
    elif name in ('AutoContrast', 'Equalize'):
        rec['op'] = _CODE[name]
    else:
        raise ValueError(f'The provided operator {name} is not recognized.')
    return rec


def normalize_batch(images_u8, ops=None, mean=IMAGE_MEAN, std=IMAGE_STD, return_bytes=False):
    """images_u8: uint8 [B,H,W,3] CUDA; ops: None or a numpy [B, num_ops] array of AUG_OP_DTYPE -> float32 [B,3,H,W]"""
    if not images_u8.is_cuda:
        raise RuntimeError('distillclip_amd has no CPU path: the augmentation kernel needs a CUDA (HIP) uint8 batch')
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[-1] != 3:
        raise ValueError(f'expected uint8 [B,H,W,3], got {images_u8.dtype} {tuple(images_u8.shape)}')
    x = images_u8.contiguous()
    B, H, W, _ = x.shape
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=x.device)
    aug = torch.empty_like(x) if return_bytes else None
    num_ops, dev_ops, ws = 0, None, None
    if ops is not None and ops.size:
        ops = np.ascontiguousarray(ops, dtype=AUG_OP_DTYPE)
        if ops.ndim != 2 or ops.shape[0] != B:
            raise ValueError(f'ops must be [B, num_ops] records, got {ops.shape}')
        num_ops = ops.shape[1]
        dev_ops = torch.from_numpy(ops.view(np.uint8).reshape(B, -1)).to(x.device, non_blocking=True)
        ws = torch.empty(lib().dclip_augment_workspace(B, H, W), dtype=torch.uint8, device=x.device)
    m = (np.ctypeslib.ctypes.c_float * 3)(*mean)
    s = (np.ctypeslib.ctypes.c_float * 3)(*std)
    lib().dclip_augment_normalize(x.data_ptr(), B, H, W, dev_ops.data_ptr() if dev_ops is not None else None, num_ops, m, s,
                                  out.data_ptr(), aug.data_ptr() if aug is not None else None,
                                  ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
                                  torch.cuda.current_stream().cuda_stream)
    return (out, aug) if return_bytes else out


class RandAugmentGPU:
    """reference rand_augment.py:90-166 RandAugment (interpolation NEAREST, fill None), batched on the GPU and fused with
    ToTensor + Normalize of ms_coco.py:19-20."""

    def __init__(self, num_ops=2, magnitude=9, num_magnitude_bins=31, mean=IMAGE_MEAN, std=IMAGE_STD):
        self.num_ops, self.magnitude, self.num_magnitude_bins = num_ops, magnitude, num_magnitude_bins
        self.mean, self.std = mean, std

    def draw(self, batch, height, width):
        """[(op name, signed magnitude)] per image, consuming torch's RNG like the reference's forward() (:152-164)"""
        meta = augmentation_space(self.num_magnitude_bins, height, width)
        names = list(meta.keys())
        plan = []
        for _ in range(batch):
            ops = []
            for _ in range(self.num_ops):
                name = names[int(torch.randint(len(meta), (1,)).item())]
                magnitudes, signed = meta[name]
                mag = float(magnitudes[self.magnitude].item()) if magnitudes.ndim > 0 else 0.0
                if signed and torch.randint(2, (1,)):
                    mag *= -1.0
                ops.append((name, mag))
            plan.append(ops)
        return plan

    def records(self, plan, height, width):
        rec = np.zeros((len(plan), self.num_ops), dtype=AUG_OP_DTYPE)
        for i, ops in enumerate(plan):
            for j, (name, mag) in enumerate(ops):
                rec[i, j] = op_record(name, mag, height, width)
        return rec

    def draw_records(self, batch, height, width, generator=None):
        """the same distribution as draw() + records() from two bulk randint calls and a 12 x 2 record table (the magnitude is
        one fixed bin, so an op and a sign determine the record); ~1000x faster on the host than the per-op loop, which is
        what keeps the input side ahead of a 13k pairs/s step.  Not the reference's RNG stream (draw() is)."""
        key = (height, width)
        if getattr(self, '_table_key', None) != key:
            meta = augmentation_space(self.num_magnitude_bins, height, width)
            table = np.zeros((len(OP_NAMES), 2), dtype=AUG_OP_DTYPE)
            for i, name in enumerate(OP_NAMES):
                magnitudes, signed = meta[name]
                mag = float(magnitudes[self.magnitude].item()) if magnitudes.ndim > 0 else 0.0
                table[i, 0] = op_record(name, mag, height, width)
                table[i, 1] = op_record(name, -mag if signed else mag, height, width)
            self._table, self._table_key = table, key
        op = torch.randint(len(OP_NAMES), (batch, self.num_ops), generator=generator).numpy()
        sign = torch.randint(2, (batch, self.num_ops), generator=generator).numpy()
        return self._table[op, sign]

    def __call__(self, images_u8, plan=None, return_bytes=False, generator=None):
        B, H, W, _ = images_u8.shape
        rec = self.records(plan, H, W) if plan is not None else self.draw_records(B, H, W, generator)
        return normalize_batch(images_u8, rec, self.mean, self.std, return_bytes)


class EvalTransformGPU:
    """validation chain of ms_coco.py:22-26 after resize / crop: ToTensor + Normalize"""

    def __init__(self, mean=IMAGE_MEAN, std=IMAGE_STD):
        self.mean, self.std = mean, std

    def __call__(self, images_u8):
        return normalize_batch(images_u8, None, self.mean, self.std)


# ---------------------------------------------------------------------------------------------------------------------
# Resize(size) + CenterCrop(size) on the GPU (reference ms_coco.py:16-17,23-24): Pillow's antialiased bilinear resample
# ---------------------------------------------------------------------------------------------------------------------
RESIZE_DESC_DTYPE = np.dtype([('src_offset', '<i8'), ('table_offset', '<i8'), ('temp_offset', '<i8'), ('height', '<i4'),
                              ('width', '<i4'), ('row0', '<i4'), ('nrows', '<i4'), ('ksize_h', '<i4'), ('ksize_v', '<i4')])
_PRECISION_BITS = 32 - 8 - 2


def _resample_coeffs(in_size, out_size):
    """libImaging Resample.c precompute_coeffs (triangle filter, support 1) + normalize_coeffs_8bpc, vectorised over the
    output index with the same double-precision operation order -> (bounds [out, 2], weights [out, ksize]) int32"""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    j = np.arange(ksize, dtype=np.int64)[None, :]
    v = np.abs((((j + xmin[:, None]) - center[:, None]) + 0.5) * ss)
    w = np.where(v < 1.0, 1.0 - v, 0.0)
    w = np.where(j < xmax[:, None], w, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for c in range(ksize):                                   # the C loop accumulates in index order
        ww = ww + w[:, c]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    kk = (0.5 + w * float(1 << _PRECISION_BITS)).astype(np.int64).astype(np.int32)
    kk = np.where(j < xmax[:, None], kk, 0).astype(np.int32)
    return np.stack([xmin, xmax], 1).astype(np.int32), kk


_TABLE_CACHE = {}


def resample_tables(height, width, size):
    """-> (int32 table block [hb | hk | vb | vk], row0, nrows, ksize_h, ksize_v) for one source size (cached)"""
    key = (height, width, size)
    hit = _TABLE_CACHE.get(key)
    if hit is not None:
        return hit
    if width <= height:                                      # torchvision Resize(int): shorter side -> size
        nw, nh = size, int(size * height / width)
    else:
        nw, nh = int(size * width / height), size
    left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
    hb, hk = _resample_coeffs(width, nw)
    vb, vk = _resample_coeffs(height, nh)
    hb, hk, vb, vk = hb[left:left + size], hk[left:left + size], vb[top:top + size].copy(), vk[top:top + size]
    row0 = int(vb[0, 0])
    nrows = int(vb[-1, 0] + vb[-1, 1]) - row0
    vb[:, 0] -= row0
    block = np.concatenate([hb.reshape(-1), hk.reshape(-1), vb.reshape(-1), vk.reshape(-1)]).astype(np.int32)
    out = (block, row0, nrows, hk.shape[1], vk.shape[1])
    if len(_TABLE_CACHE) < 4096:
        _TABLE_CACHE[key] = out
    return out


class ResizeCenterCropGPU:
    """transforms.Resize(size) + transforms.CenterCrop(size) of decoded images (list of uint8 [h, w, 3] numpy arrays or CPU
    tensors, any sizes) -> uint8 [B, size, size, 3] on the GPU, equal to the PIL result byte for byte."""

    def __init__(self, size=224):
        self.size = size
        self._pinned = None                      # grow-only pinned staging buffer (a fresh pin per batch costs more than the copy)
        self._pin_free = None                    # event after which the staging buffer may be overwritten

    def __call__(self, images, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError('distillclip_amd has no CPU path: ResizeCenterCropGPU needs a CUDA (HIP) device')
        dev = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        S, B = self.size, len(images)
        if B == 0:
            raise ValueError('ResizeCenterCropGPU: empty batch')
        desc = np.zeros(B, dtype=RESIZE_DESC_DTYPE)
        blocks, table_pos, src_pos, temp_pos = {}, 0, 0, 0
        table_list = []
        arrs = []
        for i, im in enumerate(images):
            a = im.numpy() if torch.is_tensor(im) else np.asarray(im)
            if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
                raise ValueError(f'image {i}: expected uint8 [h, w, 3], got {a.dtype} {a.shape}')
            h, w = a.shape[:2]
            if min(h, w) < 1 or max(h, w) > 16384:
                raise ValueError(f'image {i}: unsupported size {h} x {w}')
            block, row0, nrows, ksh, ksv = resample_tables(h, w, S)
            if (h, w) not in blocks:                          # images of one size share a table block
                blocks[(h, w)] = table_pos
                table_list.append(block)
                table_pos += block.size
            desc[i] = (src_pos, blocks[(h, w)], temp_pos, h, w, row0, nrows, ksh, ksv)
            arrs.append(np.ascontiguousarray(a).reshape(-1))
            src_pos += (a.size + 15) // 16 * 16
            temp_pos += (nrows * S * 3 + 15) // 16 * 16
        if self._pinned is None or self._pinned.numel() < src_pos:
            self._pinned = torch.empty(int(src_pos * 1.25) + 4096, dtype=torch.uint8).pin_memory()
        elif self._pin_free is not None:
            self._pin_free.synchronize()
        packed = self._pinned[:src_pos]
        pv = packed.numpy()
        for i, flat in enumerate(arrs):
            o = int(desc[i]['src_offset'])
            pv[o:o + flat.size] = flat
        tables = torch.from_numpy(np.concatenate(table_list))
        d_packed = packed.to(dev, non_blocking=True)
        self._pin_free = torch.cuda.Event()
        self._pin_free.record(torch.cuda.current_stream(dev))
        d_tables = tables.to(dev, non_blocking=True)
        d_desc = torch.from_numpy(desc.view(np.uint8).reshape(B, -1)).to(dev, non_blocking=True)
        ws = torch.empty(max(temp_pos, 16), dtype=torch.uint8, device=dev)
        out = torch.empty((B, S, S, 3), dtype=torch.uint8, device=dev)
        lib().dclip_resize_center_crop(d_packed.data_ptr(), d_desc.data_ptr(), d_tables.data_ptr(), B, S, out.data_ptr(),
                                       ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        for t in (d_packed, d_tables, d_desc):
            t.record_stream(torch.cuda.current_stream())
        return out
