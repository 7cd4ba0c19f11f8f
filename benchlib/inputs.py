"""Synthetic inputs of bench.py: S-scene (a 10 m x 10 m column of a scene), S-randn (the reference
benchmark's own input, benchmark.py:16-18) and S-dup25 (a quarter of the rows duplicated, dataset/semantic_dataset.py:101-106)."""
import numpy as np

def s_scene(seed, b, n):
    rs = np.random.RandomState(seed)
    xy = rs.uniform(-5, 5, (b, n, 2))
    z = np.clip(np.abs(rs.normal(0, 1.5, (b, n, 1))), 0, 8)
    rgb = rs.uniform(0, 1, (b, n, 3))
    return np.concatenate([xy, z, rgb], axis=2).astype(np.float32)


def s_randn(seed, b, n):
    """S-randn: the reference benchmark's own input (benchmark.py:16-18: np.random.randn(batch, num_point, 6))"""
    return np.random.RandomState(seed).randn(b, n, 6).astype(np.float32)


def s_dup25(seed, b, n):
    """S-dup25: S-scene with a quarter of its rows duplicates of other rows (xyz and colour), shuffled -- how the reference
    fills a cloud shorter than num_points_per_sample (dataset/semantic_dataset.py:101-106: np.random.choice of its own rows)"""
    rs = np.random.RandomState(seed + 7919)
    x = s_scene(seed, b, n)
    nd = n // 4
    for i in range(b):
        x[i, n - nd:] = x[i, rs.randint(0, n - nd, nd)]
        x[i] = x[i][rs.permutation(n)]
    return x
