"""CPU tests of the N4 host side: the numpy restatement of the reference's scene sampler against the frozen outputs of
the reference's own (lifted) methods, the label / PCD file helpers, and the voxel restatement against a dictionary
implementation."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
GOLD = os.path.join(ROOT, "tests", "golden", "dataset_sampler.npz")


def test_sampler_restatement_matches_frozen_reference_outputs():
    from make_dataset_golden import scene
    from oracle.dataset_oracle import FileDataOracle
    g = np.load(GOLD)
    for case in range(3):
        seed, n, npts, box = [int(v) for v in g["c%d_meta" % case]]
        pts, labels, colors = scene(seed, n)
        orc = FileDataOracle(pts, labels, colors, box, box)
        for k in range(4):
            np.random.seed(100 * case + k)
            draws = {}
            o = orc.sample(npts, draws)
            tag = "c%d_s%d_" % (case, k)
            for name, arr in zip(("centered", "raw", "labels", "colors"), o):
                assert np.array_equal(arr, g[tag + name]), (case, k, name)
            assert draws["center"] == int(g[tag + "center"]) and draws["count"] == int(g[tag + "count"])
    # both branches of _get_fix_sized_sample_mask are in the fixture
    counts = [(int(g["c%d_s%d_count" % (c, k)]), int(g["c%d_meta" % c][2])) for c in range(3) for k in range(4)]
    assert any(c > n for c, n in counts) and any(c <= n for c, n in counts)


def test_labels_and_pcd_roundtrip(tmp_path):
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "pcu", os.path.join(ROOT, "open3d-pointnet2-semantic3d_amd", "util", "point_cloud_util.py"))
    pcu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pcu)
    rs = np.random.RandomState(0)
    labels = rs.randint(0, 9, 1000).astype(np.int32)
    p = str(tmp_path / "a.labels")
    pcu.write_labels(p, labels)
    assert open(p).read().splitlines()[:3] == [str(int(v)) for v in labels[:3]]  # "%d\n" per label (point_cloud_util.py:60-63)
    back = pcu.load_labels(p)
    assert back.dtype == np.int32 and np.array_equal(back, labels)
    pts = rs.randn(500, 3).astype(np.float32)
    cols = rs.randint(0, 256, (500, 3)) / 255.0
    for binary in (True, False):
        q = str(tmp_path / ("b%d.pcd" % binary))
        pcu.write_point_cloud_pcd(q, pts, cols, binary=binary)
        rp, rc = pcu.read_point_cloud_pcd(q)
        assert rp.dtype == np.float64 and np.array_equal(rp, pts.astype(np.float64))
        assert np.array_equal(np.round(rc * 255), np.round(cols * 255))
    pcu.write_point_cloud_pcd(str(tmp_path / "c.pcd"), pts)
    rp, rc = pcu.read_point_cloud_pcd(str(tmp_path / "c.pcd"))
    assert np.array_equal(rp, pts.astype(np.float64)) and not rc.any()


def test_voxel_restatement_vs_dictionary():
    from oracle.dataset_oracle import voxel_down_sample
    rs = np.random.RandomState(3)
    pts = rs.uniform(0, 2, (4000, 3))
    cols = rs.uniform(0, 1, (4000, 3))
    labels = rs.randint(1, 9, 4000)
    vs = 0.25
    sp, sc, sl = voxel_down_sample(pts, cols, labels, vs)
    mb = pts.min(0) - vs * 0.5
    acc = {}
    for i, p in enumerate(pts):  # the published algorithm, literally: a map from voxel index to its members in input order
        acc.setdefault(tuple(np.floor((p - mb) / vs).astype(int)), []).append(i)
    keys = sorted(acc)
    assert len(keys) == len(sp)
    for v, k in enumerate(keys):
        ids = acc[k]
        s = np.zeros(3)
        for i in ids:
            s = s + pts[i]
        assert np.array_equal(sp[v], s / float(len(ids)))
        assert sl[v] == np.bincount(labels[ids]).argmax()
