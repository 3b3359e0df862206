"""Host-side data plumbing of the trainers (CPU): the class-balanced batch sampler the MAMC loss relies on."""
import collections

import pytest

from hawkeye_amd.data import BalancedBatchSampler, SyntheticDataset


def test_balanced_batches_have_n_classes_times_n_samples():
    labels = [i % 7 for i in range(100)]
    s = BalancedBatchSampler(labels, n_classes=3, n_samples=4, seed=1)
    batches = list(s)
    assert len(batches) == len(s) == 100 // 12
    for b in batches:
        cnt = collections.Counter(labels[i] for i in b)
        assert len(b) == 12 and len(cnt) == 3 and set(cnt.values()) == {4}
        assert len(set(b)) == 12                                   # no image twice in one batch
        assert all(type(i) is int for i in b)                       # plain ints: datasets seed generators with them


def test_balanced_sampler_is_seeded_and_rank_dependent():
    labels = [i % 5 for i in range(60)]
    a = list(BalancedBatchSampler(labels, 2, 3, seed=3, rank=0))
    b = list(BalancedBatchSampler(labels, 2, 3, seed=3, rank=0))
    c = list(BalancedBatchSampler(labels, 2, 3, seed=3, rank=1))
    assert a == b and a != c


def test_balanced_sampler_skips_small_classes_and_validates():
    labels = [0] * 10 + [1] * 10 + [2]            # class 2 has one image: cannot give 2 samples
    s = BalancedBatchSampler(labels, 2, 2)
    assert all(labels[i] != 2 for batch in s for i in batch)
    with pytest.raises(ValueError):
        BalancedBatchSampler(labels, 3, 2)


def test_synthetic_dataset_labels_agree_with_items():
    ds = SyntheticDataset(12, 8, 5, seed=2)
    assert ds.labels == [ds[i]['label'] for i in range(12)]
