import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def pn2():
    import pn2_amd
    return pn2_amd


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


# ---- deterministic synthetic inputs --------------------
def s_grid(seed, b, n, q=1024):
    """coords = multiples of 1/q in [0,1): every difference, square and 3-term sum is
    exact in fp32, so all arithmetic modes agree and exact distance ties are frequent."""
    rs = np.random.RandomState(seed)
    return (rs.randint(0, q, (b, n, 3)) / float(q)).astype(np.float32)


def s_randn(seed, b, n, c=3):
    """the reference benchmark's own input distribution (benchmark.py:16-18)."""
    return np.random.RandomState(seed).randn(b, n, c).astype(np.float32)


def s_scene(seed, b, n):
    """10 m x 10 m column standing on z=0 (dataset/semantic_dataset.py:109-121, semantic.json:17-18)."""
    rs = np.random.RandomState(seed)
    xy = rs.uniform(-5, 5, (b, n, 2))
    z = np.clip(np.abs(rs.normal(0, 1.5, (b, n, 1))), 0, 8)
    return np.concatenate([xy, z], axis=2).astype(np.float32)


def s_dup(seed, b, n, frac=0.25):
    """S-scene with `frac` of its rows duplicates of other rows, order shuffled: how the reference fills a cloud that is
    shorter than num_points_per_sample (dataset/semantic_dataset.py:101-106 appends np.random.choice of the same rows)."""
    rs = np.random.RandomState(seed + 7919)
    x = s_scene(seed, b, n)
    nd = int(n * frac)
    for i in range(b):
        src = rs.randint(0, n - nd, nd)
        x[i, n - nd:] = x[i, src]
        x[i] = x[i][rs.permutation(n)]
    return x
