"""Datasets for the trainer: an on-the-fly synthetic dataset (what the metric is
quoted on: N(0,1) 448x448 images, uniform labels - SURVEY.md section 8d), an
image-folder dataset with the reference's `label relpath` meta format
(dataset/dataset.py:22-64) and the class-balanced batch sampler.  The image
presets live in hawkeye_amd/transforms.py."""
import os

import numpy as np
import torch
from torch.utils.data import Dataset


class SyntheticDataset(Dataset):
    def __init__(self, n, image_size, num_classes, seed=0):
        self.n, self.size, self.k, self.seed = int(n), int(image_size), int(num_classes), seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        return {'img': torch.randn(3, self.size, self.size, generator=g),
                'label': int(torch.randint(0, self.k, (1,), generator=g))}

    @property
    def labels(self):
        # same draw order as __getitem__ (image first, then label) so that both agree
        out = []
        for i in range(self.n):
            g = torch.Generator().manual_seed(self.seed * 1000003 + i)
            torch.randn(3, self.size, self.size, generator=g)
            out.append(int(torch.randint(0, self.k, (1,), generator=g)))
        return out


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

    @property
    def labels(self):
        return [lab for lab, _ in self.items]

    def __getitem__(self, i):
        from PIL import Image
        lab, rel = self.items[i]
        img = Image.open(os.path.join(self.root, rel))
        if self.transform is not None:
            return {'img': self.transform(img), 'label': lab}
        from .transforms import to_float_tensor
        return {'img': to_float_tensor(img), 'label': lab}


class BalancedBatchSampler(torch.utils.data.Sampler):
    """Batches of `n_classes` distinct classes x `n_samples` images each - what the MAMC n-pairs loss needs to have
    same-class pairs in every batch (reference dataset/sampler.py:5-38, used by Examples/OSMENet.py:20-23).
    Own organisation: one shuffled queue per class that is refilled when it runs short; `seed`/`rank` make the
    stream reproducible and different on every data-parallel rank."""

    def __init__(self, labels, n_classes, n_samples, seed=0, rank=0):
        self.by_class = {}
        for idx, lab in enumerate(labels):
            self.by_class.setdefault(int(lab), []).append(idx)
        self.classes = sorted(c for c, v in self.by_class.items() if len(v) >= n_samples)
        if len(self.classes) < n_classes:
            raise ValueError(f'need {n_classes} classes with >= {n_samples} images, found {len(self.classes)}')
        self.n_classes, self.n_samples, self.total = int(n_classes), int(n_samples), len(labels)
        self.rng = np.random.RandomState(seed * 9973 + rank)
        self.queues = {c: [] for c in self.classes}

    def __len__(self):
        return self.total // (self.n_classes * self.n_samples)

    def _take(self, c):
        q = self.queues[c]
        if len(q) < self.n_samples:
            q[:] = [int(i) for i in self.rng.permutation(self.by_class[c])]
        out = q[:self.n_samples]
        del q[:self.n_samples]
        return out

    def __iter__(self):
        for _ in range(len(self)):
            batch = []
            for c in self.rng.choice(self.classes, self.n_classes, replace=False):
                batch.extend(self._take(int(c)))
            yield batch
