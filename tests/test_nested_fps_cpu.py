"""Nested farthest point sampling: the statement pn2_fps_nested (include/pn2_abi.h) relies on, checked with the CPU oracle
alone (the line-faithful restatement of tf_sampling.cu:111-176).

The SA levels sample each other's output (util/pointnet_util.py:36-37 chained by model.py:104-113).  If cloud Y is the
FPS + gather of cloud X (rows = picks in pick order) and that run met no tie before step m, then FPS(m, Y) = 0..m-1.
`oracle.fps_first_tie` is the checker of the tie record the HIP kernels emit (tests/test_ops_gpu.py holds them to it)."""
import zlib

import numpy as np
import pytest

from oracle import oracle as O

NO_TIE = 0x7FFFFFFF


def _clouds(rs, kind, b, n):
    if kind == "uniform":
        return rs.random_sample((b, n, 3)).astype(np.float32)
    if kind.startswith("lattice"):
        g = int(kind[7:])
        return (rs.randint(0, g, (b, n, 3)) / np.float32(g)).astype(np.float32)
    if kind == "dup":  # a quarter of the rows duplicated, as dataset/semantic_dataset.py:101-106 up-samples short clouds
        x = rs.random_sample((b, n, 3)).astype(np.float32)
        x[:, 3 * n // 4:] = x[:, :n - 3 * n // 4]
        for i in range(b):
            x[i] = x[i][rs.permutation(n)]
        return x
    if kind == "few":  # fewer distinct points than picks: the running maximum reaches 0 and the twins get picked
        x = rs.random_sample((b, n // 4, 3)).astype(np.float32)
        x = np.concatenate([x] * 4, axis=1)
        for i in range(b):
            x[i] = x[i][rs.permutation(x.shape[1])]
        return x
    raise KeyError(kind)


@pytest.mark.parametrize("kind", ["uniform", "lattice8", "lattice32", "lattice128", "lattice1024", "dup", "few"])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_sampling_an_fps_prefix_is_the_identity_unless_the_parent_run_tied(kind, mode):
    rs = np.random.RandomState(zlib.crc32(("%s%d" % (kind, mode)).encode()))
    x = _clouds(rs, kind, 6, 768)
    levels = [560, 200, 64, 16]  # 560 > 512: the (k mod 512, k) tie-break of the second level differs from index order
    idx = O.farthest_point_sample(levels[0], x, mode)
    record = O.fps_first_tie(levels[0], x, mode)
    cur = O.gather_point(x, idx)
    shortcuts = 0
    for m in levels[1:]:
        idx = O.farthest_point_sample(m, cur, mode)
        own = O.fps_first_tie(m, cur, mode)
        for i in range(x.shape[0]):
            if record[i] >= m:
                shortcuts += 1
                assert np.array_equal(idx[i], np.arange(m)), (kind, mode, m, i, int(record[i]))
        # what the kernels hand to the next level: the inherited record after a shortcut, the run's own otherwise
        record = np.where(record >= m, record, own)
        cur = O.gather_point(cur, idx)
    if kind in ("uniform", "dup"):
        assert shortcuts == 3 * x.shape[0]  # real-valued clouds (and coincident twins) never leave the identity
    if kind == "lattice8":
        assert shortcuts < 3 * x.shape[0]   # tie-heavy: the record must send clouds to the real sampler


def test_first_tie_record_semantics():
    # 0 -> far corner is unique, then the two remaining corners tie (strict) at step 2
    sq = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0.5, 0.5, 0]]], dtype=np.float32)
    assert O.farthest_point_sample(4, sq, 0).tolist() == [[0, 3, 1, 2]]
    assert O.fps_first_tie(4, sq, 0).tolist() == [2]
    # coincident twins only: benign ties do not count while the maximum stays positive ...
    tw = np.array([[[0, 0, 0], [4, 0, 0], [4, 0, 0], [1, 0, 0], [2.5, 0, 0]]], dtype=np.float32)
    assert O.fps_first_tie(3, tw, 0).tolist() == [NO_TIE]
    # ... but once it reaches 0 the twin is picked too: the record falls back to the first benign step
    assert O.fps_first_tie(5, tw, 0).tolist() == [1]
