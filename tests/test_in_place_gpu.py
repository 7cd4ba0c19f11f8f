"""The *_ld entry points (include/pn2_abi.h): a column block of a wider batch -- the xyz / rgb halves of point_cloud (b,n,6),
reference model.py:26-29 -- read where it lies gives the bits of the dense entry point on a copy; the inference forward then
launches no copy kernel at all."""
import numpy as np
import pytest

from conftest import s_randn, s_scene

pytestmark = pytest.mark.gpu


def _batch(seed, b, n, dev, gen=s_scene):
    import torch
    rs = np.random.RandomState(seed)
    pc = np.concatenate([gen(seed, b, n)[..., :3], rs.random_sample((b, n, 3)).astype(np.float32)], axis=2)
    return torch.from_numpy(pc).to(dev)


@pytest.mark.parametrize("n,m", [(8192, 1024), (4096, 512), (2048, 256), (300, 100), (64, 16)])
def test_sampler_reads_the_xyz_columns_in_place(pn2, cuda, n, m):
    import torch
    S = pn2.tf_ops.tf_sampling
    for gen in (s_scene, s_randn):
        pc = _batch(n + m, 5, n, cuda, gen)
        view, dense = pc[:, :, 0:3], pc[:, :, 0:3].contiguous()
        assert not view.is_contiguous()
        for mode in (2, 0, 1):
            ia, xa = S.farthest_point_sample_and_gather(m, view, arith_mode=mode)
            ib, xb = S.farthest_point_sample_and_gather(m, dense, arith_mode=mode)
            assert torch.equal(ia, ib) and torch.equal(xa, xb)
            assert torch.equal(S.fps_tie_record(xa, mode), S.fps_tie_record(xb, mode))
        # a nested level on the tagged output of an in-place run takes the shortcut like any other
        _, xa = S.farthest_point_sample_and_gather(m, view)
        _, xb = S.farthest_point_sample_and_gather(m, dense)
        a2, b2 = S.farthest_point_sample_and_gather(max(1, m // 4), xa), S.farthest_point_sample_and_gather(max(1, m // 4), xb)
        assert torch.equal(a2[0], b2[0]) and torch.equal(a2[1], b2[1])


@pytest.mark.parametrize("n,m,r,k", [(8192, 1024, 0.5, 32), (4096, 300, 0.8, 16), (5003, 256, 0.3, 64), (2048, 256, 0.5, 32)])
def test_ball_query_reads_the_cloud_in_place(pn2, cuda, n, m, r, k):
    """the LDS-grid kernel reads the strided cloud; shapes outside its range answer from a dense copy (same result either way)"""
    import torch
    for gen in (s_scene, s_randn):
        pc = _batch(n, 3, n, cuda, gen)
        view, dense = pc[:, :, 0:3], pc[:, :, 0:3].contiguous()
        q = dense[:, :m].contiguous()
        for mode in (1, 0, 2):
            ia, ca = pn2.query_ball_point(r, k, view, q, arith_mode=mode)
            ib, cb = pn2.query_ball_point(r, k, dense, q, arith_mode=mode)
            assert torch.equal(ia, ib) and torch.equal(ca, cb)


@pytest.mark.parametrize("n,m", [(8192, 1024), (1024, 256), (777, 64), (100, 3)])
def test_three_nn_reads_the_queries_in_place(pn2, cuda, n, m):
    import torch
    pc = _batch(n + 1, 4, n, cuda)
    view, dense = pc[:, :, 0:3], pc[:, :, 0:3].contiguous()
    known = dense[:, :m].contiguous()
    da, ia = pn2.three_nn(view, known)
    db, ib = pn2.three_nn(dense, known)
    assert torch.equal(ia, ib) and torch.equal(da, db)
    # a duplicated known point (equal distances: ties -> lowest index) and queries that coincide with known points
    known[:, 1] = known[:, 0]
    da, ia = pn2.three_nn(view, known)
    db, ib = pn2.three_nn(dense, known)
    assert torch.equal(ia, ib) and torch.equal(da, db)


def test_sa1_and_fp4_modules_read_the_batch_in_place(pn2, cuda):
    """pointnet_sa_module / pointnet_fp_module on the column-block views == on dense copies, bit for bit, at configs[1]'s SA1 / FP4
    shapes (the fused kernels' strided gathers) and at a small shape that falls back to copies"""
    import torch
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    for b, n, m in ((4, 8192, 1024), (2, 1024, 128)):
        pc = _batch(n, b, n, cuda)
        xv, pv = pc[:, :, 0:3], pc[:, :, 3:6]
        xd, pd = xv.contiguous(), pv.contiguous()
        tfu.set_default_store(tfu.VariableStore(device=cuda, seed=1))
        with torch.no_grad():
            kw = dict(npoint=m, radius=0.5, nsample=32, mlp=[32, 32, 64], mlp2=None, group_all=False, is_training=False,
                      bn_decay=None, scope="sa1")
            a = pu.pointnet_sa_module(xv, pv, **kw)
            bb = pu.pointnet_sa_module(xd, pd, **kw)
            for u, v in zip(a, bb):
                assert torch.equal(u, v)
            feats = torch.randn(b, m, 128, device=cuda)
            fa = pu.pointnet_fp_module(xv, a[0], pv, feats, [128, 128, 128], False, None, scope="fp4")
            fb = pu.pointnet_fp_module(xd, a[0], pd, feats, [128, 128, 128], False, None, scope="fp4")
            assert torch.equal(fa, fb)


def test_inference_forward_launches_no_copy_kernel(pn2, cuda):
    """the (B,N,6) batch is consumed in place -- a forward pass launches the library's kernels and nothing from
    at::native (torch profiler over one eager forward at configs[1]'s shapes), and equals the forward on dense slices"""
    import torch
    from torch.profiler import ProfilerActivity, profile
    tfu = pn2.util.tf_util
    hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
    B, N = 8, 8192   # (FP4 takes the hoisted three-layer chain from 65536 rows on; below, its wide-kernel path reads dense rows)
    hp.update(batch_size=B, num_point=N)
    pc = _batch(5, B, N, cuda)
    tfu.set_default_store(tfu.VariableStore(device=cuda, seed=2))
    with torch.no_grad():
        ref = pn2.model.get_sa_fp_features(pc, False, hp)[0]      # creates variables, folded weights (their own torch kernels)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            out = pn2.model.get_sa_fp_features(pc, False, hp)[0]
            torch.cuda.synchronize()
    assert torch.equal(out, ref)
    names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    assert names, "the profiler saw no device kernel"
    foreign = [nm for nm in names if "at::native" in nm or "elementwise" in nm or "Memcpy" in nm or "memcpy" in nm]
    assert not foreign, "copy / torch kernels in the inference forward: %s" % sorted(set(foreign))
    # the same network on dense slices (what every round before r05 did) gives the same bits
    xyz, rgb = pc[:, :, 0:3].contiguous(), pc[:, :, 3:6].contiguous()
    with torch.no_grad():
        dense = pn2.model.get_sa_fp_features(torch.cat([xyz, rgb], dim=2).clone(), False, hp)[0]
    assert torch.equal(out, dense)
