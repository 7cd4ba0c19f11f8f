"""GPU parity of the N4 ops: the device scene sampler with the reference's np.random draws replayed as inputs must
reproduce the frozen outputs of the reference's own methods bit for bit; the voxel down-sampling must equal the numpy
restatement exactly (float64)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
GOLD = os.path.join(ROOT, "tests", "golden", "dataset_sampler.npz")
pytestmark = pytest.mark.gpu


def test_scene_sampler_replays_reference_draws(pn2, cuda):
    import torch
    from make_dataset_golden import scene
    g = np.load(GOLD)
    for case in range(3):
        seed, n, npts, box = [int(v) for v in g["c%d_meta" % case]]
        pts, labels, colors = scene(seed, n)
        fd = pn2.dataset.SemanticFileData(points=pts, labels=labels, colors=colors, box_size_x=box, box_size_y=box, device=cuda)
        centers = np.array([int(g["c%d_s%d_center" % (case, k)]) for k in range(4)])
        cap = max(int(g["c%d_s%d_count" % (case, k)]) for k in range(4)) + 7
        masks = np.zeros((4, cap), np.uint8)
        for k in range(4):
            m = g["c%d_s%d_mask" % (case, k)]
            masks[k, :len(m)] = m
        c, r, l, col = fd.sample_batch(4, npts, center_indices=torch.from_numpy(centers).to(cuda),
                                       sample_masks=torch.from_numpy(masks).to(cuda), capacity=cap)
        fd.check_last()
        assert fd.last_cnt.cpu().tolist() == [int(g["c%d_s%d_count" % (case, k)]) for k in range(4)]
        for k in range(4):
            tag = "c%d_s%d_" % (case, k)
            assert np.array_equal(r[k].cpu().numpy(), g[tag + "raw"])
            assert np.array_equal(c[k].cpu().numpy(), g[tag + "centered"].astype(np.float32))  # the float32 the network is fed
            assert np.array_equal(l[k].cpu().numpy(), g[tag + "labels"])
            assert np.array_equal(col[k].cpu().numpy(), g[tag + "colors"].astype(np.float32))


def test_scene_sampler_device_rng_properties_and_errors(pn2, cuda):
    import torch
    from make_dataset_golden import scene
    pts, labels, colors = scene(7, 200000)
    fd = pn2.dataset.SemanticFileData(points=pts, labels=labels, colors=colors, box_size_x=10, box_size_y=10, device=cuda)
    fd.generator.manual_seed(1)
    c, r, l, col = fd.sample_batch(16, 8192, capacity=60000)
    fd.check_last()
    c, r = c.cpu().numpy(), r.cpu().numpy()
    assert c.shape == (16, 8192, 3) and c.dtype == np.float32
    for b in range(16):
        assert abs(c[b, :, 2].min()) == 0.0                       # min z == 0 (:109-121)
        assert np.isclose(c[b, :, 0].min(), -5.0) and np.isclose(c[b, :, 1].min(), -5.0)
        assert c[b, :, 0].max() <= 5.0 + 1e-6 and c[b, :, 1].max() <= 5.0 + 1e-6   # a 10 m x 10 m column
        sel = fd.last_sel[b].cpu().numpy()
        if fd.last_cnt[b] > 8192:
            assert len(np.unique(sel)) == 8192 and (np.diff(sel) > 0).all()  # a subset, scene order kept
    # a capacity that is too small is reported, never silently truncated; a wrong mask is reported
    c2 = fd.sample_batch(2, 1024, capacity=64)
    with pytest.raises(RuntimeError):
        fd.check_last()
    bad = torch.zeros((1, 60000), dtype=torch.uint8, device=cuda)
    fd.sample_batch(1, 1024, sample_masks=bad, capacity=60000)
    assert fd.last_status.cpu().tolist()[0] in (3, 0)  # 3 unless the column happened to be smaller than 1024


@pytest.mark.parametrize("n,vs", [(5000, 0.25), (100000, 0.05), (3, 1.0)])
def test_voxel_downsample_equals_restatement(pn2, cuda, n, vs):
    import torch
    from oracle.dataset_oracle import voxel_down_sample
    rs = np.random.RandomState(n)
    pts = rs.uniform(-3, 3, (n, 3)).astype(np.float32).astype(np.float64)
    cols = rs.randint(0, 256, (n, 3)) / 255.0
    labels = rs.randint(0, 9, n).astype(np.int32)
    sp, sc, sl = pn2.downsample.down_sample_arrays(torch.from_numpy(pts).to(cuda), torch.from_numpy(cols).to(cuda),
                                                   torch.from_numpy(labels).to(cuda), voxel_size=vs)
    keep = labels != 0  # downsample.py:29-43
    rp, rc, rl = voxel_down_sample(pts[keep], cols[keep], labels[keep], vs)
    assert sp.shape[0] == len(rp)
    assert np.array_equal(sp.cpu().numpy(), rp) and np.array_equal(sc.cpu().numpy(), rc)
    assert np.array_equal(sl.cpu().numpy(), rl)


def test_voxel_downsample_large_properties(pn2, cuda):
    import torch
    n, vs = 2000000, 0.05
    g = torch.Generator(device=cuda).manual_seed(0)
    pts = (torch.rand((n, 3), device=cuda, generator=g, dtype=torch.float64) * torch.tensor([20.0, 20.0, 3.0], device=cuda, dtype=torch.float64))
    labels = torch.randint(1, 9, (n,), device=cuda, generator=g, dtype=torch.int32)
    sp, sc, sl = pn2.downsample.down_sample_arrays(pts, None, labels, voxel_size=vs)
    mb = pts.min(0).values - vs * 0.5
    vox = torch.floor((pts - mb) / vs).long()
    key = (vox[:, 0] << 42) | (vox[:, 1] << 21) | vox[:, 2]
    uk, counts = torch.unique(key, return_counts=True)
    assert sp.shape[0] == uk.numel()
    # every output point lies in its own voxel and the outputs are sorted by voxel index
    ov = torch.floor((sp - mb) / vs).long()
    okey = (ov[:, 0] << 42) | (ov[:, 1] << 21) | ov[:, 2]
    assert torch.equal(okey, uk)
    # count-weighted mean of the voxel centroids == mean of the cloud
    w = counts.double()[:, None]
    assert torch.allclose((sp * w).sum(0) / n, pts.mean(0), rtol=1e-9, atol=1e-9)
    assert int(sl.min()) >= 1 and int(sl.max()) <= 8


def test_scene_sampler_default_mask_survives_tied_random_keys(pn2, cuda, monkeypatch):
    """the default random subset must select EXACTLY npts entries even when two random keys are equal (`keys <=
    k-th key` selected npts + 1, the kernel rejected the sample with status 3 and returned uninitialised memory).  Forced here
    with keys quantised to 1/64, i.e. thousands of ties per row; a rejected sample also comes back zero-filled, never garbage."""
    import torch
    from make_dataset_golden import scene
    pts, labels, colors = scene(9, 150000)
    fd = pn2.dataset.SemanticFileData(points=pts, labels=labels, colors=colors, box_size_x=10, box_size_y=10, device=cuda)
    real_rand = torch.rand
    monkeypatch.setattr(torch, "rand", lambda *a, **k: torch.floor(real_rand(*a, **k) * 64) / 64)
    c, r, l, col = fd.sample_batch(8, 4096, capacity=50000)
    monkeypatch.undo()
    assert fd.last_status.cpu().tolist() == [0] * 8
    fd.check_last()
    for b in range(8):
        if fd.last_cnt[b] > 4096:
            assert len(np.unique(fd.last_sel[b].cpu().numpy())) == 4096
    # a rejected sample (capacity too small) is zero-filled and reported; strict=True raises from sample_batch itself
    c2, r2, l2, col2 = fd.sample_batch(2, 1024, capacity=64)
    assert any(fd.last_status.cpu().tolist())
    bad = [i for i, st in enumerate(fd.last_status.cpu().tolist()) if st]
    assert float(c2[bad].abs().max()) == 0.0 and int(l2[bad].abs().max()) == 0
    fds = pn2.dataset.SemanticFileData(points=pts, labels=labels, colors=colors, box_size_x=10, box_size_y=10, device=cuda, strict=True)
    with pytest.raises(RuntimeError):
        fds.sample_batch(2, 1024, capacity=64)
