"""pn2_coarse_geometry: the sampling, ball query and 3-NN tables of the coarse levels in ONE launch, held bit for bit against
the separate entry points (which are themselves pinned to the reference's kernels in test_ref_gpu.py / test_ops_gpu.py):
farthest_point_sample + gather_point (util/pointnet_util.py:36-37 -> tf_sampling.cu:111-191), query_ball_point
(util/pointnet_util.py:39 -> tf_grouping.cu:3-43), three_nn (util/pointnet_util.py:300 -> tf_interpolate.cpp:213-243)."""
import numpy as np
import pytest

from conftest import s_dup, s_grid, s_randn, s_scene

pytestmark = pytest.mark.gpu

NO_TIE = 0x7fffffff


def T(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _cloud(gen, seed, b, n):
    if gen == "scene":
        return s_scene(seed, b, n)
    if gen == "randn":
        return s_randn(seed, b, n)
    if gen == "dup25":
        return s_dup(seed, b, n)
    if gen.startswith("grid"):
        return s_grid(seed, b, n, int(gen[4:]))
    raise ValueError(gen)


def _separate(pn2, xyz0, npoints, radii, nsamples, want_nn=True):
    """the same levels through the separate ops (tie records chained exactly as the layer API does)"""
    S = pn2.tf_ops.tf_sampling
    out, cur = [], xyz0
    for m, r, ns in zip(npoints, radii, nsamples):
        fidx, new_xyz = S.farthest_point_sample_and_gather(m, cur)
        idx, cnt = pn2.query_ball_point(r, ns, cur, new_xyz)
        nn = pn2.three_nn(cur, new_xyz) if want_nn else None
        out.append((fidx, new_xyz, idx, cnt, nn))
        cur = new_xyz
    return out


_BUILD_OF = {(2, 1): "fast_noslp", (0, 0): "off"}   # (fps mode, ball-query mode) -> the oracle/_ref build that produces it


def _assert_equals_reference_kernels(got, xyz0, npoints, radii, nsamples, build, what):
    """every level against the REFERENCE's own kernels (oracle/_ref: tf_sampling.cu / tf_grouping.cu built for gfx950), directly:
    farthestpointsamplingKernel + gatherpointKernel + query_ball_point_gpu on the level's source cloud"""
    from oracle import ref
    if not ref.available(build):
        pytest.skip("oracle/_ref not built")
    cur = xyz0.cpu().numpy()
    for l, (g, m, r, ns) in enumerate(zip(got, npoints, radii, nsamples)):
        f = ref.farthest_point_sample(m, cur, build)
        nx = ref.gather_point(cur, f, build)
        ri, rc = ref.query_ball_point(r, ns, cur, nx, build)
        assert np.array_equal(g["fps_idx"].cpu().numpy(), f), "%s level %d: picks != reference kernel" % (what, l)
        assert np.array_equal(g["new_xyz"].cpu().numpy(), nx), "%s level %d: new_xyz != reference kernel" % (what, l)
        assert np.array_equal(g["cnt"].cpu().numpy(), rc), "%s level %d: pts_cnt != reference kernel" % (what, l)
        assert np.array_equal(g["idx"].cpu().numpy(), ri), "%s level %d: ball query != reference kernel" % (what, l)
        cur = nx


def _assert_same(got, want, what):
    import torch
    for l, (g, w) in enumerate(zip(got, want)):
        assert torch.equal(g["fps_idx"], w[0]), "%s level %d: picks" % (what, l)
        assert torch.equal(g["new_xyz"], w[1]), "%s level %d: new_xyz" % (what, l)
        assert torch.equal(g["idx"], w[2]), "%s level %d: ball query idx, first bad row %s" % (
            what, l, (g["idx"] != w[2]).nonzero()[:1].tolist())
        assert torch.equal(g["cnt"], w[3]), "%s level %d: pts_cnt" % (what, l)
        if w[4] is not None:
            assert torch.equal(g["nn"][1], w[4][1]), "%s level %d: three_nn idx" % (what, l)
            assert torch.equal(g["nn"][0], w[4][0]), "%s level %d: three_nn dist" % (what, l)


@pytest.mark.parametrize("gen", ["scene", "randn", "dup25", "grid1024", "grid16", "grid4"])
def test_coarse_geometry_equals_the_separate_ops(pn2, cuda, gen):
    """configs[1]'s pyramid below the first level (1024 -> 256 -> 64 -> 16, radii 1 / 2 / 4 scaled to the cloud, K = 32):
    the source cloud is a real FPS output carrying its tie record, so real-valued clouds take the nested shortcut at every
    level and tie-heavy lattices run the in-kernel sampler -- both against the separate ops, all three contraction builds."""
    pu, S = pn2.util.pointnet_util, pn2.tf_ops.tf_sampling
    b = 16
    x = T(_cloud(gen, 3, b, 8192), cuda)
    scale = 1.0 if gen in ("scene", "dup25") else (0.5 if gen == "randn" else 0.12)
    radii = [1.0 * scale, 2.0 * scale, 4.0 * scale]
    for fps_mode, bq_mode in [(2, 1), (0, 0), (1, 2)]:
        with pn2.config.arith(fps=fps_mode, bq=bq_mode):
            _, l1 = S.farthest_point_sample_and_gather(1024, x)
            rec = S.fps_tie_record(l1)
            assert rec is not None
            want = _separate(pn2, l1, [256, 64, 16], radii, [32, 32, 32])
            got = pu.coarse_geometry(l1, [256, 64, 16], radii, [32, 32, 32])
            _assert_same(got, want, "%s modes %d/%d" % (gen, fps_mode, bq_mode))
            if (fps_mode, bq_mode) in _BUILD_OF:  # not only transitively -- the reference's kernels, level by level
                _assert_equals_reference_kernels(got, l1, [256, 64, 16], radii, [32, 32, 32], _BUILD_OF[(fps_mode, bq_mode)],
                                                 "%s modes %d/%d" % (gen, fps_mode, bq_mode))
            took = (rec >= 256).cpu().numpy()
            if gen in ("scene", "randn", "dup25"):
                assert took.all()
            if gen in ("grid16", "grid4"):
                assert not took.any()
            # the last level's record rides on its new_xyz: a further level nests on it exactly as on the separate chain
            a = S.farthest_point_sample_and_gather(8, got[-1]["new_xyz"])
            bb = S.farthest_point_sample_and_gather(8, want[-1][1])
            assert all(np.array_equal(u.cpu().numpy(), v.cpu().numpy()) for u, v in zip(a, bb))


def test_coarse_geometry_without_a_tie_record_samples_every_level(pn2, cuda):
    """an untagged source cloud (tie_in NULL): every level runs the in-kernel sampler; ragged sizes, more clouds than CUs
    would take at once, short and empty ball-query rows (tiny radius), nsample above and below the hit counts, no 3-NN."""
    pu = pn2.util.pointnet_util
    for b, n0, npoints, radii, nsamples, nn in [
            (5, 1000, [250, 77, 9], [0.8, 1.5, 3.0], [16, 48, 5], True),
            (300, 200, [50, 10], [0.05, 1e-3], [8, 4], True),
            (3, 1024, [1024, 512], [0.3, 0.6], [64, 100], False),
            (2, 64, [16, 3], [4.0, 9.0], [32, 32], True)]:
        x = T(s_scene(n0 + b, b, n0), cuda)
        want = _separate(pn2, x, npoints, radii, nsamples, nn)
        got = pu.coarse_geometry(x, npoints, radii, nsamples, want_nn=nn)
        _assert_same(got, want, "b=%d n0=%d" % (b, n0))
        assert (got[-1]["cnt"] <= nsamples[-1]).all()


def test_coarse_geometry_fuzz_against_the_separate_ops(pn2, cuda):
    """40 random pyramids (1-4 levels, ragged sizes down to 3 points, radii from empty to all-inclusive balls, nsample 1..80,
    clouds drawn from four distributions incl. heavy duplication and coarse lattices, tagged or untagged source clouds)."""
    import torch
    pu, S = pn2.util.pointnet_util, pn2.tf_ops.tf_sampling
    rs = np.random.RandomState(2024)
    for it in range(40):
        b = int(rs.choice([1, 2, 3, 7, 16, 33]))
        n0 = int(rs.randint(8, 1025))
        nlev = int(rs.randint(1, 5))
        npoints, n = [], n0
        for _ in range(nlev):
            m = int(rs.randint(3, min(n, 256) + 1))
            npoints.append(m)
            n = m
        kind = rs.choice(["scene", "randn", "dup", "grid"])
        nbig = n0 if rs.rand() < 0.4 else int(n0 * rs.uniform(1.5, 4.0))  # source = an FPS output of a bigger cloud (tagged) or raw
        if kind == "scene":
            x = s_scene(it, b, nbig)
        elif kind == "randn":
            x = s_randn(it, b, nbig)
        elif kind == "dup":
            x = s_dup(it, b, max(nbig, 8), 0.5)
        else:
            x = s_grid(it, b, nbig, int(rs.choice([2, 4, 32])))
        xt = T(x, cuda)
        if nbig > n0:
            _, src = S.farthest_point_sample_and_gather(n0, xt)
        else:
            src = xt
        ext = float(np.abs(x).max()) + 1e-3
        radii = [float(ext * rs.choice([1e-4, 0.05, 0.2, 0.6, 3.0])) for _ in range(nlev)]
        nsamples = [int(rs.choice([1, 4, 16, 32, 64, 80])) for _ in range(nlev)]
        want = _separate(pn2, src, npoints, radii, nsamples)
        got = pu.coarse_geometry(src, npoints, radii, nsamples)
        _assert_same(got, want, "fuzz %d (%s b=%d n0=%d levels=%s)" % (it, kind, b, n0, npoints))


def test_coarse_geometry_device_side_branch_per_cloud(pn2, oracle, cuda):
    """The shortcut is decided per cloud on the device: forged records make a cloud answer the identity where it is wrong
    (tie-heavy lattice) and force the sampler where the identity would have been right -- the ball query and the 3-NN table
    below follow whatever the level produced."""
    import ctypes
    import torch
    from pn2_amd import _lib
    lib, ptr = _lib.lib, _lib.ptr
    n0, m, ns, r = 1024, 256, 16, 0.2
    x = np.concatenate([s_grid(7, 2, n0, 8), s_scene(7, 2, n0)], axis=0)  # clouds 0,1 tie-heavy, 2,3 real-valued
    xt = T(x, cuda)
    fps = oracle.farthest_point_sample(m, x, 2)
    assert not np.array_equal(fps[0], np.arange(m))
    for tie_in, ident in [(None, [False] * 4), ([NO_TIE] * 4, [True] * 4), ([NO_TIE, m - 1, m, 0], [True, False, True, False])]:
        fi = torch.full((4, m), -1, dtype=torch.int32, device=cuda)
        nx = torch.zeros((4, m, 3), dtype=torch.float32, device=cuda)
        bi = torch.full((4, m, ns), -1, dtype=torch.int32, device=cuda)
        bc = torch.full((4, m), -1, dtype=torch.int32, device=cuda)
        nd = torch.zeros((4, n0, 3), dtype=torch.float32, device=cuda)
        ni = torch.full((4, n0, 3), -1, dtype=torch.int32, device=cuda)
        tout = torch.full((4,), -7, dtype=torch.int32, device=cuda)
        tin = None if tie_in is None else torch.tensor(tie_in, dtype=torch.int32, device=cuda)
        one = lambda t: (ctypes.c_void_p * 1)(t.data_ptr())  # noqa: E731
        _lib.check(lib.pn2_coarse_geometry(4, n0, 1, (ctypes.c_int * 1)(m), (ctypes.c_float * 1)(r), (ctypes.c_int * 1)(ns),
                                           ptr(xt), ptr(tin), one(fi), one(nx), one(bi), one(bc), one(nd), one(ni), ptr(tout), 2, 1,
                                           _lib.stream_ptr()), "pn2_coarse_geometry")
        for i in range(4):
            picks = np.arange(m) if ident[i] else fps[i]
            assert np.array_equal(fi[i].cpu().numpy(), picks), (tie_in, i)
            new = x[i][picks]
            assert np.array_equal(nx[i].cpu().numpy(), new)
            wi, wc = pn2.query_ball_point(r, ns, xt[i:i + 1], T(new[None], cuda))
            assert torch.equal(bi[i:i + 1], wi) and torch.equal(bc[i:i + 1], wc)
            wd, wn = pn2.three_nn(xt[i:i + 1], T(new[None], cuda))
            assert torch.equal(ni[i:i + 1], wn) and torch.equal(nd[i:i + 1], wd)
            if ident[i]:
                assert int(tout[i]) == tie_in[i]


def test_coarse_geometry_refuses_what_it_cannot_do(pn2, cuda):
    import ctypes
    import torch
    from pn2_amd import _lib
    lib, ptr = _lib.lib, _lib.ptr
    pu = pn2.util.pointnet_util
    assert pu.coarse_geometry_fits(1024, [256, 64, 16]) and not pu.coarse_geometry_fits(2048, [256])
    assert not pu.coarse_geometry_fits(1024, [512, 64]) and pu.coarse_geometry_fits(1024, [512, 64], want_nn=False)
    assert not pu.coarse_geometry_fits(100, [200]) and not pu.coarse_geometry_fits(1024, [256, 64, 16, 8, 4])
    x = T(s_scene(0, 2, 2048), cuda)
    with pytest.raises(_lib.Pn2Error):
        pu.coarse_geometry(x, [256], [1.0], [8])           # n0 > 1024
    with pytest.raises(ValueError):
        pu.coarse_geometry(x[:, :512], [64, 8], [1.0], [8, 8])
    y = x[:, :512].contiguous()
    with pytest.raises(_lib.Pn2Error):
        pu.coarse_geometry(y, [600], [1.0], [8])           # more samples than points
    with pytest.raises(_lib.Pn2Error):
        pu.coarse_geometry(y, [300], [1.0], [8])           # 3-NN table over more than 256 samples
    z = torch.empty((2, 300, 3), device=cuda)
    buf = torch.empty((2, 300, 8), dtype=torch.int32, device=cuda)
    one = lambda t: (ctypes.c_void_p * 1)(t.data_ptr())  # noqa: E731
    rc = lib.pn2_coarse_geometry(2, 512, 1, (ctypes.c_int * 1)(300), (ctypes.c_float * 1)(1.0), (ctypes.c_int * 1)(8), ptr(y), None,
                                 None, one(z), one(buf), None, None, None, None, 2, 1, _lib.stream_ptr())
    assert rc == -2  # PN2_ENULL: fps_idx is required


def test_model_forward_is_the_same_with_and_without_the_coarse_launch(pn2, cuda):
    """configs[1] forward (inference, B = 16, N = 8192): levels 2-4 and FP1-FP3's 3-NN tables from the one launch vs from the separate ops --
    the features are the same bits (every consumer reads the same indices and coordinates), and so is compute_geometry."""
    import torch
    pu = pn2.util.pointnet_util
    hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
    pc = T(np.concatenate([s_scene(1, 16, 8192), np.random.RandomState(2).random_sample((16, 8192, 3)).astype(np.float32)], 2), cuda)
    pn2.util.tf_util.set_default_store(pn2.util.tf_util.VariableStore(device=cuda, seed=0))
    outs, geos = {}, {}
    for flag in (True, False):
        pu.USE_COARSE_GEOMETRY = flag
        try:
            with torch.no_grad():
                lib = pn2._lib.lib
                lib.trace = []
                f, ep = pn2.model.get_sa_fp_features(pc, False, hp)
                names = [t[0] for t in lib.trace]
                lib.trace = None
                outs[flag] = (f.clone(), [t.clone() for t in ep["xyzs"]])
                geos[flag] = pn2.model.compute_geometry(pc[:, :, :3].contiguous(), hp)
            assert ("pn2_coarse_geometry" in names) == flag
            if flag:  # one launch instead of three samplers, three ball queries and three 3-NN searches
                assert names.count("pn2_coarse_geometry") == 1 and names.count("pn2_three_nn") == 1
                assert names.count("pn2_query_ball_point") == 1
                assert len(names) <= 22, names  # library launches per forward at configs[1]'s sizes (29 in round 3)
        finally:
            pu.USE_COARSE_GEOMETRY = True
    assert torch.equal(outs[True][0], outs[False][0])
    assert all(torch.equal(a, b) for a, b in zip(outs[True][1], outs[False][1]))
    ta, tb = pn2.model.geometry_tensors(geos[True]), pn2.model.geometry_tensors(geos[False])
    assert len(ta) == len(tb) and all(torch.equal(a, b) for a, b in zip(ta, tb))
