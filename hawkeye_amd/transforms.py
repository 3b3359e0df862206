"""Image-side of the trainers without torchvision (absent from the image): the training / evaluation presets the
reference builds in `train.py:171-183` from `dataset/transforms.py:14-73` -
    train: RandomResizedCrop(size) -> RandomHorizontalFlip(0.5) -> TrivialAugmentWide -> to float tensor ->
           Normalize(ImageNet mean / std) -> RandomErasing(p = 0.1)
    eval : Resize(resize_size) -> CenterCrop(size) -> to float tensor -> Normalize
restated over PIL + numpy + torch from the published definitions of those transforms (torchvision 0.13+ semantics:
RandomResizedCrop scale (0.08, 1) / ratio (3/4, 4/3) with ten attempts and a centre-crop fallback; TrivialAugmentWide:
one of 14 operations drawn uniformly, its strength one of 31 bins drawn uniformly, nearest-neighbour geometry, no
fill; RandomErasing scale (0.02, 0.33) / ratio (0.3, 3.3), value 0).  Host-side plumbing on the CPU data workers
(SURVEY 8f-3); randomness comes from python's `random`, which the DataLoader seeds per worker.

PARITY UNPINNED: torchvision is neither in the build container nor on the GPU box (no wheel, no network), so the
reference's presets cannot be executed to generate fixtures.  tests/test_transforms_cpu.py checks this file against the
published definitions only; the device tail `hk_image_finalize` is pinned bit-exactly against this file's own CPU path
(tests/test_gpu_kernels.py::test_image_finalize_bit_exact).
"""
import math
import random

import numpy as np
import torch
from PIL import Image, ImageEnhance, ImageOps

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
_BILINEAR, _NEAREST = Image.BILINEAR, Image.NEAREST


def to_float_tensor(img):
    """PIL RGB -> float32 [3,H,W] in [0,1] (PILToTensor + ConvertImageDtype)."""
    a = np.array(img.convert('RGB'), dtype=np.uint8)          # a writable copy
    return torch.from_numpy(a).permute(2, 0, 1).to(torch.float32).div(255)


def normalize(t, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    m = torch.tensor(mean, dtype=t.dtype).view(-1, 1, 1)
    s = torch.tensor(std, dtype=t.dtype).view(-1, 1, 1)
    return t.sub(m).div(s)


# --------------------------------------------------------------------------- geometry
def random_resized_crop_box(width, height, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), rng=random):
    """(left, top, w, h) of the crop: area fraction uniform in `scale`, aspect log-uniform in `ratio`, ten attempts,
    then the largest centre crop whose aspect is inside `ratio`."""
    area = float(width * height)
    lo, hi = math.log(ratio[0]), math.log(ratio[1])
    for _ in range(10):
        target = area * rng.uniform(scale[0], scale[1])
        aspect = math.exp(rng.uniform(lo, hi))
        w = int(round(math.sqrt(target * aspect)))
        h = int(round(math.sqrt(target / aspect)))
        if 0 < w <= width and 0 < h <= height:
            return rng.randint(0, width - w), rng.randint(0, height - h), w, h
    in_ratio = width / height
    if in_ratio < ratio[0]:
        w, h = width, int(round(width / ratio[0]))
    elif in_ratio > ratio[1]:
        w, h = int(round(height * ratio[1])), height
    else:
        w, h = width, height
    return (width - w) // 2, (height - h) // 2, w, h


def _inverse_affine(center, angle, translate, shear):
    """Coefficients (a, b, c, d, e, f) of the OUTPUT -> INPUT map PIL's Image.transform(AFFINE) wants, for a rotation by
    `angle` and shears (sx, sy) in degrees about `center` followed by `translate` (scale 1): M = T C RSS C^-1,
    RSS = [[cos(a - sy)/cos(sy), -cos(a - sy) tan(sx)/cos(sy) - sin(a)], [sin(a - sy)/cos(sy), -sin(a - sy) tan(sx)/cos(sy) + cos(a)]]."""
    rot, sx, sy = math.radians(angle), math.radians(shear[0]), math.radians(shear[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]                             # inverse of [[a, b], [c, d]] (determinant 1 at scale 1)
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


def _affine(img, translate=(0, 0), shear=(0.0, 0.0), center=None):
    """Nearest-neighbour affine warp of a PIL image; `center` in pixel coordinates (None: the middle of the image)."""
    w, h = img.size
    ctr = (w * 0.5, h * 0.5) if center is None else center
    return img.transform((w, h), Image.AFFINE, _inverse_affine(ctr, 0.0, translate, shear), _NEAREST)


# --------------------------------------------------------------------------- TrivialAugmentWide
_BINS = 31


def _ta_space():
    lin = lambda hi: np.linspace(0.0, hi, _BINS)
    return {
        'Identity': (None, False),
        'ShearX': (lin(0.99), True), 'ShearY': (lin(0.99), True),
        'TranslateX': (lin(32.0), True), 'TranslateY': (lin(32.0), True),
        'Rotate': (lin(135.0), True),
        'Brightness': (lin(0.99), True), 'Color': (lin(0.99), True), 'Contrast': (lin(0.99), True),
        'Sharpness': (lin(0.99), True),
        'Posterize': (8 - np.round(np.arange(_BINS) / ((_BINS - 1) / 6)).astype(int), False),
        'Solarize': (np.linspace(255.0, 0.0, _BINS), False),
        'AutoContrast': (None, False), 'Equalize': (None, False),
    }


TA_SPACE = _ta_space()
TA_OPS = tuple(TA_SPACE)


def apply_op(img, op, magnitude):
    """One TrivialAugmentWide operation on a PIL RGB image."""
    if op == 'Identity':
        return img
    if op == 'ShearX':       # shear about the top-left corner, as the auto-augment family does
        return _affine(img, shear=(math.degrees(math.atan(magnitude)), 0.0), center=(0.0, 0.0))
    if op == 'ShearY':
        return _affine(img, shear=(0.0, math.degrees(math.atan(magnitude))), center=(0.0, 0.0))
    if op == 'TranslateX':
        return _affine(img, translate=(int(magnitude), 0))
    if op == 'TranslateY':
        return _affine(img, translate=(0, int(magnitude)))
    if op == 'Rotate':
        return img.rotate(magnitude, _NEAREST)
    if op == 'Brightness':
        return ImageEnhance.Brightness(img).enhance(1.0 + magnitude)
    if op == 'Color':
        return ImageEnhance.Color(img).enhance(1.0 + magnitude)
    if op == 'Contrast':
        return ImageEnhance.Contrast(img).enhance(1.0 + magnitude)
    if op == 'Sharpness':
        return ImageEnhance.Sharpness(img).enhance(1.0 + magnitude)
    if op == 'Posterize':
        return ImageOps.posterize(img, int(magnitude))
    if op == 'Solarize':
        return ImageOps.solarize(img, magnitude)
    if op == 'AutoContrast':
        return ImageOps.autocontrast(img)
    if op == 'Equalize':
        return ImageOps.equalize(img)
    raise ValueError(f'unknown operation {op}')


def trivial_augment_wide(img, rng=random):
    op = TA_OPS[rng.randrange(len(TA_OPS))]
    mags, signed = TA_SPACE[op]
    magnitude = 0.0 if mags is None else float(mags[rng.randrange(_BINS)])
    if signed and rng.randrange(2):
        magnitude = -magnitude
    return apply_op(img, op, magnitude)


# --------------------------------------------------------------------------- erasing
def random_erasing_box(height, width, scale=(0.02, 0.33), ratio=(0.3, 3.3), rng=random):
    """(top, left, h, w) of the erased rectangle or None when ten draws do not fit."""
    area = height * width
    lo, hi = math.log(ratio[0]), math.log(ratio[1])
    for _ in range(10):
        erase = area * rng.uniform(scale[0], scale[1])
        aspect = math.exp(rng.uniform(lo, hi))
        h, w = int(round(math.sqrt(erase * aspect))), int(round(math.sqrt(erase / aspect)))
        if h < height and w < width:
            return rng.randint(0, height - h), rng.randint(0, width - w), h, w
    return None


# --------------------------------------------------------------------------- presets
class ClassificationPresetTrain:
    """dataset/transforms.py:14-50 as instantiated by train.py:173-177."""

    def __init__(self, crop_size, mean=IMAGENET_MEAN, std=IMAGENET_STD, hflip_prob=0.5, auto_augment_policy='ta_wide',
                 random_erase_prob=0.1, device_finalize=False):
        """device_finalize: stop after the PIL stages and return {'u8': uint8 [H,W,3], 'erase': int32 [4]}; the float
        conversion, normalisation and erasing then run on the GPU (hawkeye_amd.functional.image_finalize) - same
        numbers, a quarter of the host->device bytes."""
        if auto_augment_policy not in (None, 'ta_wide'):
            raise ValueError('only the policy the reference trainers use (ta_wide) is provided')
        self.size, self.mean, self.std = int(crop_size), mean, std
        self.hflip_prob, self.policy, self.erase_prob = hflip_prob, auto_augment_policy, random_erase_prob
        self.device_finalize = device_finalize

    def __call__(self, img):
        img = img.convert('RGB')
        left, top, w, h = random_resized_crop_box(*img.size)
        img = img.crop((left, top, left + w, top + h)).resize((self.size, self.size), _BILINEAR)
        if self.hflip_prob > 0 and random.random() < self.hflip_prob:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        if self.policy == 'ta_wide':
            img = trivial_augment_wide(img)
        box = None
        if self.erase_prob > 0 and random.random() < self.erase_prob:
            box = random_erasing_box(self.size, self.size)
        if self.device_finalize:
            return {'u8': torch.from_numpy(np.array(img, dtype=np.uint8)),
                    'erase': torch.tensor(box if box is not None else (0, 0, 0, 0), dtype=torch.int32)}
        t = normalize(to_float_tensor(img), self.mean, self.std)
        if box is not None:
            top, left, h, w = box
            t[:, top:top + h, left:left + w] = 0.0
        return t


class ClassificationPresetEval:
    """dataset/transforms.py:53-73: shorter side to `resize_size`, centre crop, normalise."""

    def __init__(self, crop_size, resize_size=256, mean=IMAGENET_MEAN, std=IMAGENET_STD, device_finalize=False):
        self.size, self.resize, self.mean, self.std = int(crop_size), int(resize_size), mean, std
        self.device_finalize = device_finalize

    def __call__(self, img):
        img = img.convert('RGB')
        w, h = img.size
        if w <= h:
            nw, nh = self.resize, int(self.resize * h / w)
        else:
            nw, nh = int(self.resize * w / h), self.resize
        img = img.resize((nw, nh), _BILINEAR)
        left, top = int(round((nw - self.size) / 2.0)), int(round((nh - self.size) / 2.0))
        if nw < self.size or nh < self.size:                      # pad with zeros like CenterCrop does for small images
            canvas = Image.new('RGB', (max(nw, self.size), max(nh, self.size)))
            canvas.paste(img, ((canvas.size[0] - nw) // 2, (canvas.size[1] - nh) // 2))
            img, (nw, nh) = canvas, canvas.size
            left, top = int(round((nw - self.size) / 2.0)), int(round((nh - self.size) / 2.0))
        img = img.crop((left, top, left + self.size, top + self.size))
        if self.device_finalize:
            return {'u8': torch.from_numpy(np.array(img, dtype=np.uint8)), 'erase': torch.zeros(4, dtype=torch.int32)}
        return normalize(to_float_tensor(img), self.mean, self.std)
