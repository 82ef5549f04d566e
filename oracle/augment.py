"""ORACLE (test infrastructure only): the reference's training image transform after decoding,
    Resize(224) -> CenterCrop(224) -> RandAugment(num_ops=4) -> ToTensor -> Normalize(CLIP mean / std)
(reference data/component/ms_coco.py:15-26, rand_augment.py:10-87 `_apply_op`, :128-166 RandAugment, utils.py:11-12).

The reference runs these on PIL images through torchvision.transforms.functional.  torchvision is absent from this image
(and unpinned by the reference); Pillow (12.2.0 here) is present.  Two layers:
  * `pil_*`  — torchvision's published PIL code path (functional.py affine / rotate / adjust_* / posterize / autocontrast /
               equalize are thin wrappers over Image.transform, ImageEnhance, ImageOps) restated over the real Pillow calls;
               this is the pinned behaviour: the pixel arithmetic is Pillow's own.
  * `np_*`   — a numpy restatement of Pillow's pixel algorithms (fixed-point nearest affine, Image.blend, SMOOTH filter,
               L conversion, histogram LUTs), which is what the HIP kernel implements; tests/test_augment_cpu.py checks
               np_* == pil_* bit for bit over every op and magnitude bin.
Parity status: pinned against Pillow 12.2.0 (third-party dependency of the reference), torchvision formulas restated.
"""
import math

import numpy as np

OPS = ['Identity', 'ShearX', 'ShearY', 'TranslateX', 'TranslateY', 'Rotate', 'Brightness', 'Contrast', 'Sharpness',
       'Posterize', 'AutoContrast', 'Equalize']          # key order of rand_augment.py:128-143 (_augmentation_space)
SIGNED = {'ShearX', 'ShearY', 'TranslateX', 'TranslateY', 'Rotate', 'Brightness', 'Contrast', 'Sharpness'}
IMAGE_MEAN = (0.48145466, 0.4578275, 0.40821073)         # reference data/component/utils.py:11-12
IMAGE_STD = (0.26862954, 0.26130258, 0.27577711)


def magnitude_of(op, bin_index, height, width, num_bins=31):
    """rand_augment.py:128-143: the magnitude table entry (unsigned) of `op` at `bin_index`"""
    import torch
    if op in ('Identity', 'AutoContrast', 'Equalize'):
        return 0.0
    if op in ('ShearX', 'ShearY'):
        return float(torch.linspace(0.0, 0.3, num_bins)[bin_index])
    if op == 'TranslateX':
        return float(torch.linspace(0.0, 150.0 / 331.0 * width, num_bins)[bin_index])
    if op == 'TranslateY':
        return float(torch.linspace(0.0, 150.0 / 331.0 * height, num_bins)[bin_index])
    if op == 'Rotate':
        return float(torch.linspace(0.0, 30.0, num_bins)[bin_index])
    if op in ('Brightness', 'Contrast', 'Sharpness'):
        return float(torch.linspace(0.0, 0.9, num_bins)[bin_index])
    if op == 'Posterize':
        return float((8 - (torch.arange(num_bins) / ((num_bins - 1) / 4)).round().int())[bin_index])
    raise ValueError(op)


# ---------------------------------------------------------------------------------------------------------------------
# torchvision.transforms.functional._get_inverse_affine_matrix (published formula; RSS = rotate-scale-shear)
# ---------------------------------------------------------------------------------------------------------------------
def inverse_affine_matrix(center, angle, translate, scale, shear):
    rot = math.radians(angle)
    sx, sy = math.radians(shear[0]), math.radians(shear[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m = [x / scale for x in m]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


def rotate_matrix(angle, w, h):
    """PIL.Image.Image.rotate's matrix (expand=False, center=None, translate=None) for angles off the 0/90/180/270 fast paths"""
    angle = angle % 360.0
    center = (w / 2, h / 2)
    ang = -math.radians(angle)
    m = [round(math.cos(ang), 15), round(math.sin(ang), 15), 0.0, round(-math.sin(ang), 15), round(math.cos(ang), 15), 0.0]
    m[2] = m[0] * -center[0] + m[1] * -center[1] + m[2]
    m[5] = m[3] * -center[0] + m[4] * -center[1] + m[5]
    m[2] += center[0]
    m[5] += center[1]
    return m


def op_matrix(op, magnitude, w, h):
    """the 6 affine coefficients the reference hands to Image.transform for a geometric op (rand_augment.py:13-66)"""
    if op == 'ShearX':
        return inverse_affine_matrix([0, 0], 0.0, [0, 0], 1.0, [math.degrees(math.atan(magnitude)), 0.0])
    if op == 'ShearY':
        return inverse_affine_matrix([0, 0], 0.0, [0, 0], 1.0, [0.0, math.degrees(math.atan(magnitude))])
    if op == 'TranslateX':
        return inverse_affine_matrix([w * 0.5, h * 0.5], 0.0, [int(magnitude), 0], 1.0, [0.0, 0.0])
    if op == 'TranslateY':
        return inverse_affine_matrix([w * 0.5, h * 0.5], 0.0, [0, int(magnitude)], 1.0, [0.0, 0.0])
    if op == 'Rotate':
        return rotate_matrix(magnitude, w, h)
    raise ValueError(op)


# ---------------------------------------------------------------------------------------------------------------------
# layer 1: the real Pillow calls behind torchvision's functional_pil
# ---------------------------------------------------------------------------------------------------------------------
def pil_apply_op(img, op, magnitude):
    from PIL import Image, ImageEnhance, ImageOps
    w, h = img.size
    if op == 'Identity':
        return img
    if op in ('ShearX', 'ShearY', 'TranslateX', 'TranslateY'):
        return img.transform((w, h), Image.AFFINE, op_matrix(op, magnitude, w, h), Image.NEAREST, fillcolor=(0, 0, 0))
    if op == 'Rotate':
        return img.rotate(magnitude, Image.NEAREST, False, None, fillcolor=(0, 0, 0))
    if op == 'Brightness':
        return ImageEnhance.Brightness(img).enhance(1.0 + magnitude)
    if op == 'Contrast':
        return ImageEnhance.Contrast(img).enhance(1.0 + magnitude)
    if op == 'Sharpness':
        return ImageEnhance.Sharpness(img).enhance(1.0 + magnitude)
    if op == 'Posterize':
        return ImageOps.posterize(img, int(magnitude))
    if op == 'AutoContrast':
        return ImageOps.autocontrast(img)
    if op == 'Equalize':
        return ImageOps.equalize(img)
    raise ValueError(op)


def pil_rand_augment(arr, ops):
    """arr: uint8 [H, W, 3]; ops: [(name, signed magnitude)] already drawn -> uint8 [H, W, 3]"""
    from PIL import Image
    img = Image.fromarray(arr, 'RGB')
    for name, mag in ops:
        img = pil_apply_op(img, name, mag)
    return np.asarray(img).copy()


def to_tensor_normalize(arr, mean=IMAGE_MEAN, std=IMAGE_STD):
    """transforms.ToTensor + Normalize on uint8 [H, W, 3] -> float32 [3, H, W] (torch float32 arithmetic)"""
    import torch
    t = torch.from_numpy(arr).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    m = torch.as_tensor(mean, dtype=torch.float32).view(-1, 1, 1)
    s = torch.as_tensor(std, dtype=torch.float32).view(-1, 1, 1)
    return t.sub_(m).div_(s).numpy()


def pil_resize_center_crop(img, size=224):
    """transforms.Resize(size) (shorter side, bilinear, antialias) + CenterCrop(size) on a PIL image"""
    from PIL import Image
    w, h = img.size
    if w <= h:
        nw, nh = size, int(size * h / w)
    else:
        nh, nw = size, int(size * w / h)
    if (w, h) != (nw, nh):
        img = img.resize((nw, nh), Image.BILINEAR)
    left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
    return img.crop((left, top, left + size, top + size))


# ---------------------------------------------------------------------------------------------------------------------
# layer 2: numpy restatement of Pillow's pixel algorithms (what the HIP kernel does)
# ---------------------------------------------------------------------------------------------------------------------
def _fix(v):
    return int(math.floor(v * 65536.0 + 0.5))


def affine_fixed_coeffs(m):
    """libImaging Geometry.c affine_fixed: 16.16 fixed-point coefficients, half-pixel centre folded into the offsets"""
    a0, a1, a3, a4 = _fix(m[0]), _fix(m[1]), _fix(m[3]), _fix(m[4])
    a2 = _fix(m[2] + m[0] * 0.5 + m[1] * 0.5)
    a5 = _fix(m[5] + m[3] * 0.5 + m[4] * 0.5)
    return a0, a1, a2, a3, a4, a5


def np_affine_nearest(arr, m):
    h, w = arr.shape[:2]
    if m[1] == 0 and m[3] == 0:
        return np_scale_nearest(arr, m)
    a0, a1, a2, a3, a4, a5 = affine_fixed_coeffs(m)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.int64)
    xx = a2 + a1 * ys + a0 * xs
    yy = a5 + a4 * ys + a3 * xs
    xin, yin = xx >> 16, yy >> 16
    ok = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
    out = np.zeros_like(arr)
    out[ok] = arr[yin[ok], xin[ok]]
    return out


def np_scale_nearest(arr, m):
    """libImaging Geometry.c ImagingScaleAffine (taken when a[1] == a[3] == 0: the translate ops): double-precision source
    coordinates truncated towards zero, columns / rows outside the source stay at the fill colour"""
    h, w = arr.shape[:2]
    out = np.zeros_like(arr)
    xo = m[2] + m[0] * 0.5
    yo = m[5] + m[4] * 0.5
    xin = np.full(w, -1, dtype=np.int64)
    for x in range(w):
        xi = int(xo) if xo >= 0 else -1          # COORD(v) = v < 0 ? -1 : (int) v
        if 0 <= xi < w:
            xin[x] = xi
        xo += m[0]
    cols = np.nonzero(xin >= 0)[0]
    for y in range(h):
        yi = int(yo) if yo >= 0 else -1
        if 0 <= yi < h and len(cols):
            out[y, cols] = arr[yi, xin[cols]]
        yo += m[4]
    return out


def np_blend(deg, img, factor):
    """libImaging Blend.c: out = in1 + alpha (in2 - in1) in float; truncation inside [0, 1], clip + truncation outside"""
    a = np.float32(factor)
    d = deg.astype(np.int32)
    t = (d.astype(np.float32) + a * (img.astype(np.int32) - d).astype(np.float32)).astype(np.float32)
    if 0.0 <= factor <= 1.0:
        return t.astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def np_luma(arr):
    """libImaging Convert.c rgb2l: ITU-R 601-2 in 16.16 fixed point"""
    r, g, b = (arr[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def np_brightness(arr, factor):
    return np_blend(np.zeros_like(arr), arr, factor)


def np_contrast(arr, factor):
    l = np_luma(arr)
    mean = int(float(l.astype(np.int64).sum()) / l.size + 0.5)
    return np_blend(np.full_like(arr, mean), arr, factor)


def np_smooth(arr):
    """ImageFilter.SMOOTH = 3x3 (1 1 1 / 1 5 1 / 1 1 1) / 13: libImaging Filter.c, one-pixel border copied"""
    h, w = arr.shape[:2]
    k = (np.array([1, 1, 1, 1, 5, 1, 1, 1, 1], dtype=np.float32) / np.float32(13)).astype(np.float32)
    a = arr.astype(np.float32)
    out = arr.copy()
    ss = np.full((h - 2, w - 2, arr.shape[2]), np.float32(0.5), dtype=np.float32)
    # rows are taken bottom row first (in1 = y + 1), each row's three taps summed before it is added
    for ky, dy in ((0, 1), (1, 0), (2, -1)):
        row = (a[1 + dy:h - 1 + dy, 0:w - 2] * k[ky * 3 + 0] + a[1 + dy:h - 1 + dy, 1:w - 1] * k[ky * 3 + 1]) + a[1 + dy:h - 1 + dy, 2:w] * k[ky * 3 + 2]
        ss = (ss + row.astype(np.float32)).astype(np.float32)
    out[1:h - 1, 1:w - 1] = np.clip(ss.astype(np.int32), 0, 255).astype(np.uint8)
    return out


def np_sharpness(arr, factor):
    return np_blend(np_smooth(arr), arr, factor)


def np_posterize(arr, bits):
    return arr & np.uint8(~(2 ** (8 - bits) - 1) & 0xFF)


def autocontrast_lut(hist):
    nz = np.nonzero(hist)[0]
    lut = np.arange(256)
    if len(nz) == 0:
        return lut.astype(np.uint8)
    lo, hi = int(nz[0]), int(nz[-1])
    if hi <= lo:
        return lut.astype(np.uint8)
    scale = 255.0 / (hi - lo)
    offset = -lo * scale
    out = np.empty(256, dtype=np.int64)
    for ix in range(256):
        v = int(ix * scale + offset)
        out[ix] = 0 if v < 0 else 255 if v > 255 else v
    return out.astype(np.uint8)


def equalize_lut(hist):
    histo = [int(f) for f in hist if f]
    if len(histo) <= 1:
        return np.arange(256).astype(np.uint8)
    step = (sum(histo) - histo[-1]) // 255
    if not step:
        return np.arange(256).astype(np.uint8)
    n = step // 2
    lut = np.empty(256, dtype=np.int64)
    for i in range(256):
        lut[i] = n // step
        n += int(hist[i])
    return np.minimum(lut, 255).astype(np.uint8)          # Image.point clips the table entries to 8 bits


def np_lut_op(arr, lut_fn):
    out = np.empty_like(arr)
    for c in range(arr.shape[2]):
        hist = np.bincount(arr[..., c].reshape(-1), minlength=256)
        out[..., c] = lut_fn(hist)[arr[..., c]]
    return out


def np_apply_op(arr, op, magnitude):
    h, w = arr.shape[:2]
    if op == 'Identity':
        return arr
    if op in ('ShearX', 'ShearY', 'TranslateX', 'TranslateY', 'Rotate'):
        if op == 'Rotate' and magnitude % 360.0 == 0:
            return arr.copy()
        return np_affine_nearest(arr, op_matrix(op, magnitude, w, h))
    if op == 'Brightness':
        return np_brightness(arr, 1.0 + magnitude)
    if op == 'Contrast':
        return np_contrast(arr, 1.0 + magnitude)
    if op == 'Sharpness':
        return np_sharpness(arr, 1.0 + magnitude)
    if op == 'Posterize':
        return np_posterize(arr, int(magnitude))
    if op == 'AutoContrast':
        return np_lut_op(arr, autocontrast_lut)
    if op == 'Equalize':
        return np_lut_op(arr, equalize_lut)
    raise ValueError(op)


def np_rand_augment(arr, ops):
    for name, mag in ops:
        arr = np_apply_op(arr, name, mag)
    return arr


# ---------------------------------------------------------------------------------------------------------------------
# Resize(size) + CenterCrop(size): numpy restatement of Pillow's two-pass antialiased bilinear resample (libImaging Resample.c:
# precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc), restricted to the crop window
# ---------------------------------------------------------------------------------------------------------------------
PRECISION_BITS = 32 - 8 - 2


def resize_target(w, h, size):
    """torchvision Resize(int): shorter side -> size, longer side int(size * long / short)"""
    if w <= h:
        return size, int(size * h / w)
    return int(size * w / h), size


def resample_coeffs(in_size, out_size):
    """-> (bounds int [out, 2] = (first source index, count), coefficients int [out, ksize]) of the triangle filter"""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        w = np.empty(xmax, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            v = -v if v < 0 else v
            w[x] = 1.0 - v if v < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w = w / ww
        for x in range(xmax):
            p = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + p) if w[x] < 0 else int(0.5 + p)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def np_resize_center_crop(arr, size=224):
    """uint8 [h, w, 3] -> uint8 [size, size, 3] == np.asarray(pil_resize_center_crop(Image.fromarray(arr), size))"""
    h, w = arr.shape[:2]
    nw, nh = resize_target(w, h, size)
    left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
    hb, hk = resample_coeffs(w, nw)
    vb, vk = resample_coeffs(h, nh)
    hb, hk, vb, vk = hb[left:left + size], hk[left:left + size], vb[top:top + size], vk[top:top + size]
    row0, row1 = int(vb[0, 0]), int(vb[-1, 0] + vb[-1, 1])
    src = arr.astype(np.int64)
    half = 1 << (PRECISION_BITS - 1)
    if w != nw:
        temp = np.empty((row1 - row0, size, 3), dtype=np.int64)
        for ox in range(size):
            x0, n = int(hb[ox, 0]), int(hb[ox, 1])
            acc = half + (src[row0:row1, x0:x0 + n, :] * hk[ox, :n].astype(np.int64)[None, :, None]).sum(axis=1)
            temp[:, ox, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
    else:
        temp = src[row0:row1, left:left + size, :]
    if h != nh:
        out = np.empty((size, size, 3), dtype=np.int64)
        for oy in range(size):
            y0, n = int(vb[oy, 0]) - row0, int(vb[oy, 1])
            acc = half + (temp[y0:y0 + n] * vk[oy, :n].astype(np.int64)[:, None, None]).sum(axis=0)
            out[oy] = np.clip(acc >> PRECISION_BITS, 0, 255)
    else:
        out = temp[top - row0:top - row0 + size]
    return out.astype(np.uint8)
