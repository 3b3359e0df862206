"""Fixtures for the host image presets from the REFERENCE's own dataset/transforms.py - runs only where
torchvision is installed (it is not in the build image: there the presets stay 'parity unpinned').
    python oracle/gen_transform_fixtures.py        # writes tests/golden/transforms_tv.npz
Test infrastructure: nothing under hawkeye_amd/ imports this."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get('HAWKEYE_REFERENCE', '/root/reference')

try:
    import torchvision  # noqa: F401
except ImportError:
    sys.exit('torchvision is not installed here: nothing generated (the presets stay parity-unpinned)')
sys.path.insert(0, REF)
from PIL import Image  # noqa: E402
from dataset.transforms import ClassificationPresetEval  # noqa: E402  (reference dataset/transforms.py:36-73)

crop, resize = 224, 256
out = {'n': 3, 'crop': crop, 'resize': resize}
for k, (w, h) in enumerate(((500, 375), (333, 500), (256, 256))):
    img = (np.random.RandomState(40 + k).rand(h, w, 3) * 255).astype(np.uint8)
    out[f'img{k}'] = img
    out[f'eval{k}'] = ClassificationPresetEval(crop_size=crop, resize_size=resize)(Image.fromarray(img)).numpy()
np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'transforms_tv.npz'), **out)
print('wrote tests/golden/transforms_tv.npz')
