"""Host-side image presets (hawkeye_amd/transforms.py): torchvision is not in the image, so these check
the published semantics of the transforms on synthetic images rather than a reference run."""
import math
import random

import numpy as np
import pytest
import torch
from PIL import Image

from hawkeye_amd import transforms as T


def _img(w=40, h=30, seed=0):
    return Image.fromarray((np.random.RandomState(seed).rand(h, w, 3) * 255).astype(np.uint8))


def _arr(img):
    return np.asarray(img, dtype=np.int32)


def test_random_resized_crop_box_statistics_and_fallback():
    rng = random.Random(1)
    fracs, aspects = [], []
    for _ in range(4000):
        l, t, w, h = T.random_resized_crop_box(400, 300, rng=rng)
        assert 0 <= l and 0 <= t and l + w <= 400 and t + h <= 300 and w > 0 and h > 0
        fracs.append(w * h / 120000.0)
        aspects.append(w / h)
    assert 0.06 < min(fracs) < 0.12 and max(fracs) <= 1.0 and 0.38 < float(np.mean(fracs)) < 0.56    # large draws with a wrong aspect are rejected
    assert min(aspects) > 0.70 and max(aspects) < 1.40
    # an extreme panorama never fits a 3/4..4/3 crop of >= 8 % area in ten draws often -> the fallback centre crop
    l, t, w, h = T.random_resized_crop_box(2000, 10, scale=(0.9, 1.0), rng=random.Random(0))
    assert h == 10 and w == int(round(10 * 4 / 3)) and l == (2000 - w) // 2 and t == 0


def test_translate_and_shear_geometry():
    img = _img()
    a = _arr(img)
    tx = _arr(T.apply_op(img, 'TranslateX', 5.0))
    assert (tx[:, 5:] == a[:, :-5]).all() and (tx[:, :5] == 0).all()
    ty = _arr(T.apply_op(img, 'TranslateY', -4.0))
    assert (ty[:-4] == a[4:]).all() and (ty[-4:] == 0).all()
    # ShearX by m about the top-left corner: output(x, y) samples input(x + m' y, y); row 0 is unchanged
    sh = _arr(T.apply_op(img, 'ShearX', 0.5))
    assert (sh[0] == a[0]).all()
    shift = [int(np.argmax((sh[y] != 0).any(axis=1))) for y in (8, 16, 24)]
    assert shift[0] < shift[1] < shift[2] or shift[0] > shift[1] > shift[2] or shift == [0, 0, 0]
    sy = _arr(T.apply_op(img, 'ShearY', 0.5))
    assert (sy[:, 0] == a[:, 0]).all()


def test_colour_and_bit_operations():
    img = _img(seed=3)
    a = _arr(img)
    assert T.apply_op(img, 'Identity', 0.0) is img
    assert (_arr(T.apply_op(img, 'Posterize', 2)) == (a & 0xC0)).all()
    sol = _arr(T.apply_op(img, 'Solarize', 100.0))
    assert (sol == np.where(a < 100, a, 255 - a)).all()
    assert (_arr(T.apply_op(img, 'Brightness', 0.0)) == a).all()
    assert _arr(T.apply_op(img, 'Brightness', 0.9)).mean() > a.mean() > _arr(T.apply_op(img, 'Brightness', -0.9)).mean()
    grey = _arr(T.apply_op(img, 'Color', -1.0))
    assert (grey[..., 0] == grey[..., 1]).all() and (grey[..., 1] == grey[..., 2]).all()
    ac = _arr(T.apply_op(Image.fromarray((a // 2 + 40).astype(np.uint8)), 'AutoContrast', 0.0))
    assert ac.min() == 0 and ac.max() == 255
    r = _arr(T.apply_op(img, 'Rotate', 90.0))
    assert r.shape == a.shape


def test_trivial_augment_space_and_sampling():
    assert len(T.TA_OPS) == 14
    assert T.TA_SPACE['Posterize'][0].tolist()[0] == 8 and T.TA_SPACE['Posterize'][0].tolist()[-1] == 2
    assert T.TA_SPACE['Solarize'][0][0] == 255.0 and T.TA_SPACE['Solarize'][0][-1] == 0.0
    assert math.isclose(T.TA_SPACE['Rotate'][0][-1], 135.0) and math.isclose(T.TA_SPACE['ShearX'][0][-1], 0.99)
    random.seed(5)
    img = _img(64, 64)
    seen = set()
    for _ in range(300):
        out = T.trivial_augment_wide(img)
        assert out.size == img.size and out.mode == 'RGB'
    rng = random.Random(2)
    for _ in range(300):
        seen.add(T.TA_OPS[rng.randrange(14)])
    assert len(seen) == 14


def test_presets_shapes_determinism_and_erasing():
    img = _img(500, 375, seed=7)
    tr = T.ClassificationPresetTrain(224)
    random.seed(11)
    a = tr(img)
    random.seed(11)
    b = tr(img)
    assert a.shape == (3, 224, 224) and a.dtype == torch.float32 and torch.equal(a, b)
    ev = T.ClassificationPresetEval(224, 256)
    e = ev(img)
    assert e.shape == (3, 224, 224) and torch.equal(e, ev(img))
    # eval = resize shorter side to 256 then centre crop: the centre pixel survives (up to resampling)
    small = T.ClassificationPresetEval(8, 4)(_img(4, 4))
    assert small.shape == (3, 8, 8)
    # normalisation constants
    flat = Image.new('RGB', (300, 300), (124, 116, 104))         # ~ ImageNet mean * 255
    z = T.ClassificationPresetEval(224, 256)(flat)
    assert z.abs().max() < 0.02
    # erasing boxes: inside the image, area fraction within the published range
    rng = random.Random(3)
    for _ in range(500):
        box = T.random_erasing_box(224, 224, rng=rng)
        assert box is not None
        top, left, h, w = box
        assert top + h <= 224 and left + w <= 224 and 0.015 < h * w / 224.0 ** 2 < 0.34
    always = T.ClassificationPresetTrain(64, auto_augment_policy=None, random_erase_prob=1.0)
    random.seed(1)
    t = always(_img(200, 200))
    assert (t == 0).any()
    with pytest.raises(ValueError):
        T.ClassificationPresetTrain(64, auto_augment_policy='ra')
