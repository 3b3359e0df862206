"""Meters / accuracy / timer / seeding used by the trainer (behaviour of the
reference's utils/utils.py:10-108, minus its 1-second sleep per log record, :76)."""
import logging
import random
import time

import numpy as np
import torch


class AverageMeter:
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


class PerformanceMeter:
    """Tracks current and best value of a metric."""

    def __init__(self, higher_is_better=True):
        self.best_function = max if higher_is_better else min
        self.reset()

    def reset(self):
        self.values = []

    def update(self, new_value):
        self.values.append(new_value)

    @property
    def current_value(self):
        return self.values[-1]

    value = current_value

    @property
    def best_value(self):
        return self.best_function(self.values)

    @property
    def best_epoch(self):
        return self.values.index(self.best_value)


def accuracy(scores, targets, k):
    """top-k accuracy in percent (utils/utils.py:52-66); one device->host read."""
    _, ind = scores.topk(k, 1, True, True)
    correct = ind.eq(targets.view(-1, 1).expand_as(ind))
    return correct.view(-1).float().sum().item() * (100.0 / targets.size(0))


class Timer:
    def __init__(self):
        self.start = self.t = time.time()

    def tick(self):
        now = time.time()
        d, self.t = now - self.t, now
        return d


def set_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


class TqdmHandler(logging.StreamHandler):
    """Stream handler that does not break tqdm bars (no sleep)."""

    def emit(self, record):
        try:
            from tqdm import tqdm
            tqdm.write(self.format(record))
        except Exception:
            super().emit(record)


class ScalarWriter:
    """tensorboardX.SummaryWriter stand-in (not installed here): uses tensorboardX when importable,
    else appends `tag,step,value` lines to <log_root>/scalars.csv."""

    def __init__(self, log_root):
        self._tb, self._f = None, None
        try:
            from tensorboardX import SummaryWriter
            self._tb = SummaryWriter(log_root)
        except Exception:
            import os
            self._f = open(os.path.join(log_root, 'scalars.csv'), 'a')

    def add_scalar(self, tag, value, step):
        if self._tb is not None:
            self._tb.add_scalar(tag, value, step)
        else:
            self._f.write(f'{tag},{step},{value}\n')
            self._f.flush()

    def close(self):
        if self._tb is not None:
            self._tb.close()
        elif self._f is not None:
            self._f.close()
