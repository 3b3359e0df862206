"""Datasets for the trainer: an on-the-fly synthetic dataset (what the metric is
quoted on: N(0,1) 448x448 images, uniform labels - SURVEY.md section 8d) and a
minimal image-folder dataset with the reference's `label relpath` meta format
(dataset/dataset.py:22-64) using PIL + torch ops only (torchvision is absent).
The reference's augmentation stack (TrivialAugmentWide, RandomErasing, mixup) is
out of scope for the hot path."""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

MEAN = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
STD = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)


class SyntheticDataset(Dataset):
    def __init__(self, n, image_size, num_classes, seed=0):
        self.n, self.size, self.k, self.seed = int(n), int(image_size), int(num_classes), seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        return {'img': torch.randn(3, self.size, self.size, generator=g),
                'label': int(torch.randint(0, self.k, (1,), generator=g))}


def _to_tensor(img):
    a = np.asarray(img.convert('RGB'), dtype=np.uint8)
    return torch.from_numpy(a).permute(2, 0, 1).float().div_(255.0)


class TrainTransform:
    """random-resized-crop (scale 0.35-1) + horizontal flip + normalise"""

    def __init__(self, crop_size):
        self.crop = crop_size

    def __call__(self, img):
        w, h = img.size
        area = w * h * float(np.random.uniform(0.35, 1.0))
        ratio = float(np.exp(np.random.uniform(np.log(3 / 4), np.log(4 / 3))))
        cw, ch = min(w, int(round((area * ratio) ** 0.5))), min(h, int(round((area / ratio) ** 0.5)))
        x0, y0 = np.random.randint(0, w - cw + 1), np.random.randint(0, h - ch + 1)
        img = img.crop((x0, y0, x0 + cw, y0 + ch)).resize((self.crop, self.crop), resample=2)
        t = _to_tensor(img)
        if np.random.rand() < 0.5:
            t = t.flip(2)
        return (t - MEAN) / STD


class EvalTransform:
    """resize shorter side + centre crop + normalise"""

    def __init__(self, crop_size, resize_size):
        self.crop, self.resize = crop_size, resize_size

    def __call__(self, img):
        w, h = img.size
        s = self.resize / min(w, h)
        img = img.resize((max(self.crop, int(round(w * s))), max(self.crop, int(round(h * s)))), resample=2)
        w, h = img.size
        x0, y0 = (w - self.crop) // 2, (h - self.crop) // 2
        return (_to_tensor(img.crop((x0, y0, x0 + self.crop, y0 + self.crop))) - MEAN) / STD


class FGDataset(Dataset):
    """meta file lines: `<label> <relative/path.jpg>` (metadata/cub/train.txt:1)."""

    def __init__(self, root, meta_path, transform=None):
        self.root, self.transform, self.items = root, transform, []
        with open(meta_path) as f:
            for line in f:
                line = line.strip()
                if line:
                    lab, rel = line.split(' ', 1)
                    self.items.append((int(lab), rel))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        from PIL import Image
        lab, rel = self.items[i]
        img = Image.open(os.path.join(self.root, rel))
        return {'img': self.transform(img) if self.transform else _to_tensor(img), 'label': lab}
