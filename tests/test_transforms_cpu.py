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


class _ScriptedRng:
    """random.Random stand-in that returns scripted values: uniform(a, b) -> a + u (b - a) for the next scripted u in
    [0, 1]; randint(a, b) / randrange(n) -> the next scripted integer (checked against the range)."""

    def __init__(self, uniforms=(), ints=()):
        self.u, self.i = list(uniforms), list(ints)

    def uniform(self, a, b):
        return a + self.u.pop(0) * (b - a)

    def randint(self, a, b):
        v = self.i.pop(0)
        assert a <= v <= b, (a, v, b)
        return v

    def randrange(self, n):
        v = self.i.pop(0)
        assert 0 <= v < n, (v, n)
        return v


def test_random_resized_crop_known_answers():
    """Hand-computed answers of torchvision's RandomResizedCrop.get_params arithmetic (the preset of
    dataset/transforms.py:26): target area = A u1 with u1 uniform in scale, aspect = exp(uniform(log 3/4, log 4/3)),
    w = round(sqrt(area * aspect)), h = round(sqrt(area / aspect)), accepted when it fits, offsets uniform integers."""
    # 400 x 300 image, area fraction 0.08 + 0.5 * 0.92 = 0.54 -> 64 800 px; log-aspect at the middle of its range = 0 -> 1:1
    # -> w = h = round(254.558) = 255, fits (255 <= 300)
    box = T.random_resized_crop_box(400, 300, rng=_ScriptedRng(uniforms=[0.5, 0.5], ints=[100, 7]))
    assert box == (100, 7, 255, 255)
    # aspect at the upper end (4/3), area fraction 1.0: w = round(sqrt(120000 * 4/3)) = 400, h = round(sqrt(90000)) = 300
    box = T.random_resized_crop_box(400, 300, rng=_ScriptedRng(uniforms=[1.0, 1.0], ints=[0, 0]))
    assert box == (0, 0, 400, 300)
    # first draw does not fit (fraction 1.0, aspect 3/4: w = 300, h = 400 > 300), second does (fraction 0.08, aspect 3/4:
    # area 9600 -> w = round(sqrt(7200)) = 85, h = round(sqrt(12800)) = 113)
    box = T.random_resized_crop_box(400, 300, rng=_ScriptedRng(uniforms=[1.0, 0.0, 0.0, 0.0], ints=[315, 187]))
    assert box == (315, 187, 85, 113)
    # ten misses -> centre crop at the clamped aspect: 1000 x 100 image (ratio 10 > 4/3) -> w = round(100 * 4/3) = 133
    box = T.random_resized_crop_box(1000, 100, scale=(1.0, 1.0), rng=_ScriptedRng(uniforms=[0.0, 0.0] * 10))
    assert box == ((1000 - 133) // 2, 0, 133, 100)
    # erasing (dataset/transforms.py:32: RandomErasing defaults): 224 x 224, fraction 0.02 + 0.5 * 0.31 = 0.175 -> 8780.8 px,
    # log-aspect mid-range: ratio (0.3, 3.3) -> exp((ln 0.3 + ln 3.3) / 2) = sqrt(0.99) -> h = round(sqrt(8780.8 * 0.994987))
    # = round(93.47) = 93, w = round(sqrt(8780.8 / 0.994987)) = round(93.94) = 94
    assert T.random_erasing_box(224, 224, rng=_ScriptedRng(uniforms=[0.5, 0.5], ints=[10, 20])) == (10, 20, 93, 94)


def test_trivial_augment_wide_op_table_known_answers():
    """TrivialAugmentWide's augmentation space with 31 bins (torchvision autoaugment: _augmentation_space of
    TrivialAugmentWide), written out: the op list in its order, the magnitude of selected bins, the whole Posterize
    table (8 - round(k / 5)), which ops are signed - and that a scripted draw (op, bin, sign) applies exactly that."""
    assert T.TA_OPS == ('Identity', 'ShearX', 'ShearY', 'TranslateX', 'TranslateY', 'Rotate', 'Brightness', 'Color',
                        'Contrast', 'Sharpness', 'Posterize', 'Solarize', 'AutoContrast', 'Equalize')
    signed = {op for op in T.TA_OPS if T.TA_SPACE[op][1]}
    assert signed == {'ShearX', 'ShearY', 'TranslateX', 'TranslateY', 'Rotate', 'Brightness', 'Color', 'Contrast', 'Sharpness'}
    assert T.TA_SPACE['Posterize'][0].tolist() == [8] * 3 + [7] * 5 + [6] * 5 + [5] * 5 + [4] * 5 + [3] * 5 + [2] * 3
    for op, top in (('ShearX', 0.99), ('TranslateY', 32.0), ('Rotate', 135.0), ('Contrast', 0.99)):
        m = T.TA_SPACE[op][0]
        assert len(m) == 31 and m[0] == 0.0 and math.isclose(m[15], top / 2) and math.isclose(m[30], top)
        assert math.isclose(m[7], top * 7 / 30)
    sol = T.TA_SPACE['Solarize'][0]
    assert sol[0] == 255.0 and math.isclose(sol[10], 255.0 * 20 / 30) and sol[30] == 0.0
    for op in ('Identity', 'AutoContrast', 'Equalize'):
        assert T.TA_SPACE[op][0] is None
    img = _img(48, 48, seed=9)
    # op 3 = TranslateX, bin 15 -> 16 px, sign draw 1 -> negative: content moves 16 px to the left
    out = T.trivial_augment_wide(img, rng=_ScriptedRng(ints=[3, 15, 1]))
    assert (_arr(out)[:, :-16] == _arr(img)[:, 16:]).all() and (_arr(out)[:, -16:] == 0).all()
    # op 10 = Posterize, bin 30 -> 2 bits kept (unsigned op: no sign draw)
    out = T.trivial_augment_wide(img, rng=_ScriptedRng(ints=[10, 30]))
    assert (_arr(out) == (_arr(img) & 0xC0)).all()
    # op 11 = Solarize, bin 15 -> threshold 127.5
    out = T.trivial_augment_wide(img, rng=_ScriptedRng(ints=[11, 15]))
    assert (_arr(out) == np.where(_arr(img) < 127.5, _arr(img), 255 - _arr(img))).all()


def test_preset_tail_known_answers(monkeypatch):
    """Hand-derived answers for the rest of the preset parameter paths (dataset/transforms.py:26-46,63-69), labelled for what
    they are: checks against torchvision's PUBLISHED definitions, not against torchvision itself (unpinned: DESIGN.md
    section 4).  PILToTensor + ConvertImageDtype = value / 255; Normalize = (v - mean_c) / std_c per channel; the erase
    box is zeroed AFTER the normalisation (RandomErasing value = 0 on the normalised tensor); RandomHorizontalFlip mirrors
    the columns; Resize(int) takes the SHORTER side to `resize_size` and the longer to int(size * long / short) (truncated);
    CenterCrop's offsets are int(round((side - crop) / 2))."""
    # normalisation constants on a constant image: (200 / 255 - mean) / std per channel
    img = Image.fromarray(np.full((8, 8, 3), 200, dtype=np.uint8))
    out = T.normalize(T.to_float_tensor(img))
    want = [(200.0 / 255.0 - m) / s_ for m, s_ in zip((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))]
    assert out.shape == (3, 8, 8)
    for c in range(3):
        assert math.isclose(float(out[c, 3, 5]), want[c], rel_tol=1e-6)
    assert math.isclose(want[0], 1.3070474, rel_tol=1e-6) and math.isclose(want[2], 1.6813944, rel_tol=1e-6)   # (by hand)
    # black / white pixels: -mean / std and (1 - mean) / std
    bw = np.zeros((2, 2, 3), dtype=np.uint8)
    bw[0, 0] = 255
    o2 = T.normalize(T.to_float_tensor(Image.fromarray(bw)))
    assert math.isclose(float(o2[1, 1, 1]), -0.456 / 0.224, rel_tol=1e-6) and math.isclose(float(o2[1, 0, 0]), 0.544 / 0.224, rel_tol=1e-6)
    # eval preset geometry: 500 x 375 (w x h), resize 510, crop 448 -> shorter side (h) 510, w = int(510 * 500 / 375) = 680;
    # offsets int(round((680 - 448) / 2)) = 116 and int(round((510 - 448) / 2)) = 31.  A ramp image whose value encodes
    # its column makes the crop window visible.
    ramp = np.tile((np.arange(500) // 2).astype(np.uint8)[None, :, None], (375, 1, 3))
    ev = T.ClassificationPresetEval(448, 510, device_finalize=True)(Image.fromarray(ramp))['u8'].numpy()
    assert ev.shape == (448, 448, 3)
    # column x of the crop = column 116 + x of the 680-wide resize = source column ~ (116 + x + 0.5) * 500 / 680 - 0.5
    for x in (0, 200, 447):
        src = (116 + x + 0.5) * 500.0 / 680.0 - 0.5
        assert abs(float(ev[100, x, 0]) - src / 2.0) <= 1.0, (x, ev[100, x, 0], src / 2.0)
    # portrait: 300 x 400 -> w is the shorter side: 510 x int(510 * 400 / 300) = 510 x 680; left 31, top 116
    ramp_v = np.tile((np.arange(400) // 2).astype(np.uint8)[:, None, None], (1, 300, 3))
    ev = T.ClassificationPresetEval(448, 510, device_finalize=True)(Image.fromarray(ramp_v))['u8'].numpy()
    for y in (0, 447):
        src = (116 + y + 0.5) * 400.0 / 680.0 - 0.5
        assert abs(float(ev[y, 10, 0]) - src / 2.0) <= 1.0
    # train preset with every draw scripted: the crop box (whole image), a flip, no augmentation, an erase box.
    # 64 x 64 ramp -> crop (0, 0, 64, 64) resized to 32: column x holds source columns 2x, 2x + 1; flipped: 63 - ...
    rampc = np.tile((np.arange(64) * 4).astype(np.uint8)[None, :, None], (64, 1, 3))
    monkeypatch.setattr(T, 'random_resized_crop_box', lambda w, h: (0, 0, w, h))
    monkeypatch.setattr(T, 'random_erasing_box', lambda h, w: (4, 6, 5, 7))            # (top, left, h, w)
    draws = iter([0.2, 0.05])                               # flip (0.2 < 0.5), erase (0.05 < 0.1)
    monkeypatch.setattr(T.random, 'random', lambda: next(draws))
    tr = T.ClassificationPresetTrain(32, auto_augment_policy=None, random_erase_prob=0.1)(Image.fromarray(rampc))
    assert tr.shape == (3, 32, 32)
    assert float(tr[:, 4:9, 6:13].abs().max()) == 0.0       # the erased box: zeros in NORMALISED space
    px = lambda v, c: (v / 255.0 - (0.485, 0.456, 0.406)[c]) / (0.229, 0.224, 0.225)[c]
    # flipped ramp: output column x shows source columns 63 - 2x, 62 - 2x (mean 62.5 - 2x, times 4 grey levels per column)
    for x in (0, 20, 31):
        got = float(tr[1, 20, x])
        assert abs(got - px((62.5 - 2 * x) * 4.0, 1)) <= 3.0 / 255.0 / 0.224, (x, got)
    assert float(tr[1, 20, 0]) > float(tr[1, 20, 31])       # mirrored: bright on the left


def test_presets_against_torchvision_fixtures_when_present():
    """tests/golden/transforms_tv.npz is written by oracle/gen_transform_fixtures.py wherever torchvision is installed (it
    is not in this image nor on the GPU box: DESIGN.md section 4, 'parity unpinned'); when the file exists the eval preset
    must reproduce torchvision's output on the same images."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'transforms_tv.npz')
    if not os.path.exists(path):
        pytest.skip('no torchvision fixtures (oracle/gen_transform_fixtures.py needs torchvision)')
    g = np.load(path)
    for k in range(int(g['n'])):
        img = Image.fromarray(g[f'img{k}'])
        out = T.ClassificationPresetEval(int(g['crop']), int(g['resize']))(img)
        assert float((out - torch.from_numpy(g[f'eval{k}'])).abs().max()) < 2.5 / 255 / 0.224      # resampling: <= 2 grey levels
