"""oracle/_ref parity: the REFERENCE's own kernels (tf_ops/tf_sampling.cu, tf_ops/tf_grouping.cu compiled
unmodified for gfx950, oracle/Makefile `_ref`) == the C restatement (oracle/pn2_oracle.c) == the HIP kernels
of this package, bit for bit, at configs[0] and at every SA level of configs[1] (all 16 scenes).

What each _ref build pins (found by reading the ISA, asserted here):
    off         every mul/add rounded                      == oracle / HIP arithmetic mode 0
    fast_noslp  LLVM DAG-combine contraction               == mode 1 in query_ball_point_gpu
                                                              (fma(dz,dz,fma(dx,dx,dy*dy))), but mode 2 in
                                                              farthestpointsamplingKernel
                                                              (fma(dz,dz,fma(dy,dy,dx*dx))): the SAME source
                                                              expression contracts in two different orders in
                                                              the two kernels -- the order is a compiler
                                                              artefact, which is why it is an ABI parameter.
    fast        hipcc default (SLP packs two squares)      == oracle mode 3 (FPS) / 5 (ball query); the HIP
                                                              kernels do not offer these (not an nvcc form).
On S-grid inputs every operation is exact, all builds and all modes must agree.
"""
import numpy as np
import pytest

from conftest import s_dup, s_grid, s_randn, s_scene

pytestmark = pytest.mark.gpu

FPS_MODE = {"off": 0, "fast_noslp": 2, "fast": 3}
BQ_MODE = {"off": 0, "fast_noslp": 1, "fast": 5}
HIP_MODES = (0, 1, 2)
SA_LEVELS = [(1024, 0.5, 32), (256, 1.0, 32), (64, 2.0, 32), (16, 4.0, 32)]  # semantic.json:23-37


@pytest.fixture(scope="module")
def ref(cuda):
    from oracle import ref as R
    for b in R.BUILDS:
        assert R.available(b), "oracle/_ref/%s missing: `make -C oracle _ref` (needs /root/reference)" % b
    return R


@pytest.fixture(autouse=True)
def _arith_scope(pn2):
    """set_mode(m): both index kernels in arithmetic mode m for the rest of THIS test (a thread-local `config.arith` scope,
    closed when the test ends) -- the product has no process-global mode switch.  set_mode(None) = the defaults."""
    import contextlib
    global set_mode
    stack = contextlib.ExitStack()

    def set_mode(mode=None, fps=None, bq=None):
        stack.close()
        if mode is not None or fps is not None or bq is not None:
            stack.enter_context(pn2.config.arith(mode, fps=fps, bq=bq))
    yield
    stack.close()


def T(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _gen(name, seed, b, n):
    return {"grid": lambda: s_grid(seed, b, n, 1024), "scene": lambda: s_scene(seed, b, n),
            "randn": lambda: s_randn(seed, b, n)}[name]()


def _hip_fps(pn2, cuda, m, x, mode):
    set_mode(mode)
    return pn2.farthest_point_sample(m, T(x, cuda)).cpu().numpy()


def _hip_bq(pn2, cuda, r, k, x1, x2, mode):
    set_mode(mode)
    idx, cnt = pn2.query_ball_point(r, k, T(x1, cuda), T(x2, cuda))
    return idx.cpu().numpy(), cnt.cpu().numpy()


def _three_way_fps(pn2, oracle, ref, cuda, m, x, exact):
    """exact=True: grid input, every build/mode must give the same picks."""
    base = None
    for build in ref.BUILDS:
        r = ref.farthest_point_sample(m, x, build)
        o = oracle.farthest_point_sample(m, x, FPS_MODE[build])
        assert np.array_equal(r, o), "_ref[%s] != oracle mode %d at %s" % (build, FPS_MODE[build],
                                                                            np.argwhere(r != o)[:3])
        if FPS_MODE[build] in HIP_MODES:
            h = _hip_fps(pn2, cuda, m, x, FPS_MODE[build])
            assert np.array_equal(r, h), "_ref[%s] != HIP mode %d at %s" % (build, FPS_MODE[build],
                                                                             np.argwhere(r != h)[:3])
        if exact:
            base = r if base is None else base
            assert np.array_equal(r, base)
    if exact:
        for mode in HIP_MODES:
            assert np.array_equal(_hip_fps(pn2, cuda, m, x, mode), base)
    return base


def _three_way_bq(pn2, oracle, ref, cuda, r_, k, x1, x2, exact):
    base = None
    for build in ref.BUILDS:
        ri, rc = ref.query_ball_point(r_, k, x1, x2, build)
        oi, oc = oracle.query_ball_point(r_, k, x1, x2, BQ_MODE[build])
        assert np.array_equal(rc, oc) and np.array_equal(ri, oi), "_ref[%s] != oracle mode %d" % (build, BQ_MODE[build])
        if BQ_MODE[build] in HIP_MODES:
            hi, hc = _hip_bq(pn2, cuda, r_, k, x1, x2, BQ_MODE[build])
            assert np.array_equal(rc, hc) and np.array_equal(ri, hi), "_ref[%s] != HIP mode %d" % (build, BQ_MODE[build])
        if exact:
            base = (ri, rc) if base is None else base
            assert np.array_equal(ri, base[0]) and np.array_equal(rc, base[1])
    if exact:
        for mode in HIP_MODES:
            hi, hc = _hip_bq(pn2, cuda, r_, k, x1, x2, mode)
            assert np.array_equal(hi, base[0]) and np.array_equal(hc, base[1])


# ---------------------------------------------------------------- configs[0] -------------------------------
@pytest.mark.parametrize("gen", ["grid", "scene", "randn", "uniform"])
def test_config0_fps_gather_ball_group(pn2, oracle, ref, cuda, gen):
    """BASELINE configs[0]: B=2, N=1024, npoint=256, K=16, C=3, r=0.2 (U(0,1)^3 is the 'uniform' case)."""
    if gen == "uniform":
        x = np.random.RandomState(7).random_sample((2, 1024, 3)).astype(np.float32)
    else:
        x = _gen(gen, 21, 2, 1024)
    r_ = {"grid": 0.2, "scene": 0.8, "randn": 0.4, "uniform": 0.2}[gen]
    _three_way_fps(pn2, oracle, ref, cuda, 256, x, gen == "grid")
    f = ref.farthest_point_sample(256, x, "off")
    new_xyz = ref.gather_point(x, f, "off")
    assert np.array_equal(new_xyz, oracle.gather_point(x, f))
    assert np.array_equal(new_xyz, pn2.gather_point(T(x, cuda), T(f, cuda)).cpu().numpy())
    _three_way_bq(pn2, oracle, ref, cuda, r_, 16, x, new_xyz, gen == "grid")
    idx, _ = ref.query_ball_point(r_, 16, x, new_xyz, "off")
    g = ref.group_point(x, idx, "off")
    assert np.array_equal(g, oracle.group_point(x, idx))
    assert np.array_equal(g, pn2.group_point(T(x, cuda), T(idx, cuda)).cpu().numpy())


# ---------------------------------------------------------------- configs[1], every SA level, all 16 scenes -
@pytest.mark.parametrize("gen", ["grid", "scene", "randn"])
def test_config1_every_sa_level_all_scenes(pn2, oracle, ref, cuda, gen):
    """semantic.json: N=8192 -> 1024 -> 256 -> 64 -> 16, K=32, r=0.5/1/2/4, B=16.  Each level's input is the
    previous level's reference output, so every level is held to the reference on identical inputs."""
    x = _gen(gen, 0, 16, 8192)
    scale = {"grid": 0.1, "scene": 1.0, "randn": 0.6}[gen]  # grid cloud lives in [0,1)^3
    for npoint, radius, k in SA_LEVELS:
        _three_way_fps(pn2, oracle, ref, cuda, npoint, x, gen == "grid")
        f = ref.farthest_point_sample(npoint, x, "off")
        new_xyz = ref.gather_point(x, f, "off")
        assert np.array_equal(new_xyz, oracle.gather_point(x, f))
        _three_way_bq(pn2, oracle, ref, cuda, radius * scale, k, x, new_xyz, gen == "grid")
        # the fused entry point the layer API uses (FPS + gather in one launch) against the reference pair
        set_mode(0)
        hf, hxyz = pn2.tf_ops.tf_sampling.farthest_point_sample_and_gather(npoint, T(x, cuda))
        assert np.array_equal(hf.cpu().numpy(), f) and np.array_equal(hxyz.cpu().numpy(), new_xyz)
        x = new_xyz


@pytest.mark.parametrize("gen", ["scene", "randn"])
def test_default_configuration_is_the_contracted_reference_build(pn2, ref, cuda, gen):
    """the configuration the product ships, bench.py times and smoke() checks -- every op on its DEFAULT
    arithmetic mode -- equals ONE complete build of the reference's own kernels: oracle/_ref "fast_noslp" (contraction on, as
    under nvcc's --fmad=true; FPS contracts as mode 2, ball query as mode 1).  The PRODUCT's geometry chain
    (model.compute_geometry: FPS + gather + ball query of the four SA levels of semantic.json, each level fed by the
    previous level's HIP output, all 16 scenes) against the same chain run on that build, bit for bit."""
    assert (pn2.config.FPS_ARITH_DEFAULT, pn2.config.BQ_ARITH_DEFAULT) == (FPS_MODE["fast_noslp"], BQ_MODE["fast_noslp"])
    x = _gen(gen, 3, 16, 8192)
    hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
    set_mode(None)  # the defaults, nothing scoped
    geo = pn2.model.compute_geometry(T(x, cuda), hp)
    cur = x
    for li, (npoint, radius, k) in enumerate(SA_LEVELS):
        f = ref.farthest_point_sample(npoint, cur, "fast_noslp")
        new_xyz = ref.gather_point(cur, f, "fast_noslp")
        ri, rc = ref.query_ball_point(radius, k, cur, new_xyz, "fast_noslp")
        assert np.array_equal(geo["xyzs"][li + 1].cpu().numpy(), new_xyz), "level %d: sampled coordinates" % (li + 1)
        assert np.array_equal(geo["idxs"][li].cpu().numpy(), ri), "level %d: ball query indices" % (li + 1)
        cur = new_xyz
    # and the contraction-off build is what `arith(0)` selects for the same chain
    with pn2.config.arith(0):
        geo0 = pn2.model.compute_geometry(T(x, cuda), hp)
    cur = x
    for li, (npoint, radius, k) in enumerate(SA_LEVELS):
        f = ref.farthest_point_sample(npoint, cur, "off")
        new_xyz = ref.gather_point(cur, f, "off")
        ri, _ = ref.query_ball_point(radius, k, cur, new_xyz, "off")
        assert np.array_equal(geo0["xyzs"][li + 1].cpu().numpy(), new_xyz) and np.array_equal(geo0["idxs"][li].cpu().numpy(), ri)
        cur = new_xyz


NO_TIE = 0x7FFFFFFF


def _nested_chain(pn2, cuda, x, levels, mode):
    """the product's chain: every level samples the TENSOR the previous level returned (tagged with its tie record)"""
    S = pn2.tf_ops.tf_sampling
    set_mode(fps=mode)
    cur, out = T(x, cuda), []
    for m in levels:
        idx, new_xyz = S.farthest_point_sample_and_gather(m, cur)
        tie = S.fps_tie_record(new_xyz)
        assert tie is not None
        out.append((idx.cpu().numpy(), new_xyz.cpu().numpy(), tie.cpu().numpy()))
        cur = new_xyz
    return out


@pytest.mark.parametrize("gen,n,levels", [
    ("scene", 8192, (1024, 256, 64, 16)), ("randn", 8192, (1024, 256, 64, 16)), ("dup25", 8192, (1024, 256, 64, 16)),
    ("grid1024", 8192, (1024, 256, 64, 16)), ("grid16", 8192, (1024, 256, 64, 16)), ("grid4", 4096, (1024, 256, 64, 16)),
    ("scene", 1024, (256, 64, 16)), ("grid64", 2048, (600, 256, 64)), ("dup25", 700, (300, 100, 30)),
    ("few", 3000, (1024, 600, 128))])
def test_nested_fps_chain_equals_the_reference_chain(pn2, oracle, ref, cuda, gen, n, levels):
    """VERDICT r03 #1.  pn2_fps_nested: a level whose input is the previous level's output answers idx = 0..m-1 without
    sampling when the parent run met no tie before step m, and samples otherwise -- either way the reference's own kernel
    (tf_sampling.cu:111-176, oracle/_ref) run level by level on the same clouds gives the same bits.  Held here: the
    chain (both contraction builds the HIP kernels offer), the tie record against the oracle's analysis of the run, and
    WHICH path each cloud took: real-valued clouds and clouds with duplicated rows (dataset/semantic_dataset.py:101-106)
    take the shortcut at every nested level, tie-heavy lattices are sampled."""
    b = 16 if n >= 4096 else 5
    if gen == "few":  # 3000 rows, 750 distinct: the running maximum reaches 0 inside the first level
        base = s_scene(11, b, n // 4)
        x = np.concatenate([base] * 4, axis=1)
        rs = np.random.RandomState(5)
        for i in range(b):
            x[i] = x[i][rs.permutation(n)]
    elif gen == "dup25":
        x = s_dup(2, b, n)
    elif gen.startswith("grid"):
        x = s_grid(n, b, n, int(gen[4:]))
    else:
        x = _gen(gen, 1, b, n)
    for build in ("off", "fast_noslp"):
        mode = FPS_MODE[build]
        got = _nested_chain(pn2, cuda, x, levels, mode)
        cur, record, took = x, None, []
        for li, m in enumerate(levels):
            f = ref.farthest_point_sample(m, cur, build)
            new_xyz = ref.gather_point(cur, f, build)
            own = oracle.fps_first_tie(m, cur, mode)
            assert np.array_equal(got[li][0], f), "%s level %d (m=%d, %s): picks differ at %s" % (
                gen, li, m, build, np.argwhere(got[li][0] != f)[:3])
            assert np.array_equal(got[li][1], new_xyz)
            if record is None:
                expect = own
            else:
                short = record >= m
                took.append(short)
                assert np.array_equal(f[short], np.broadcast_to(np.arange(m, dtype=np.int32), f[short].shape))
                expect = np.where(short, record, own)
            # the record: exact up to step m-2 (the one-pick kernels do not look at the last pick; nobody asks for m picks of m)
            assert np.array_equal(np.minimum(got[li][2], m - 1), np.minimum(expect, m - 1)), (gen, li, build, got[li][2], expect)
            record, cur = got[li][2], new_xyz
        took = np.stack(took)
        if gen in ("scene", "randn", "dup25"):
            assert took.all(), "a real-valued cloud left the shortcut"
        if gen in ("grid16", "grid4", "few"):
            assert not took[0].any(), "a tie-heavy cloud took the shortcut at the first nested level"


def test_nested_fps_device_side_branch(pn2, oracle, cuda):
    """The choice is made per cloud ON THE DEVICE from tie_in (one graph node either way): a forged record makes the kernel
    answer the identity where it is wrong, a zero record forces the sampler where the identity would have been right."""
    import torch
    from pn2_amd import _lib
    lib, ptr = _lib.lib, _lib.ptr
    for n, m in [(1024, 256), (256, 64), (4096, 700), (8192, 1024)]:
        x = np.concatenate([s_grid(n, 2, n, 8), s_scene(n, 2, n)], axis=0)   # clouds 0,1 tie-heavy, 2,3 real-valued
        want = oracle.farthest_point_sample(m, x, 2)
        xt = T(x, cuda)
        for tie_in, ident in [(None, [False] * 4), ([NO_TIE] * 4, [True] * 4), ([0] * 4, [False] * 4),
                              ([NO_TIE, m - 1, m, 0], [True, False, True, False])]:
            out = torch.full((4, m), -1, dtype=torch.int32, device=cuda)
            nx = torch.zeros((4, m, 3), dtype=torch.float32, device=cuda)
            tout = torch.full((4,), -7, dtype=torch.int32, device=cuda)
            tin = None if tie_in is None else torch.tensor(tie_in, dtype=torch.int32, device=cuda)
            _lib.check(lib.pn2_fps_nested(4, n, m, ptr(xt), None, ptr(out), ptr(nx), ptr(tin), ptr(tout), 2,
                                          _lib.stream_ptr()), "pn2_fps_nested")
            o, t = out.cpu().numpy(), tout.cpu().numpy()
            for i in range(4):
                if ident[i]:
                    assert np.array_equal(o[i], np.arange(m)) and t[i] == tie_in[i]
                    assert np.array_equal(nx[i].cpu().numpy(), x[i, :m])
                else:
                    assert np.array_equal(o[i], want[i])
                    assert np.array_equal(nx[i].cpu().numpy(), x[i][want[i]])
        assert not np.array_equal(want[0], np.arange(m))  # the forged identity above really was a different answer


def test_nested_fps_edge_shapes(pn2, oracle, cuda):
    """pn2_fps_nested at the edges: more clouds than the reference's 32 blocks, m == n (every point picked: the record must
    not send the next level to the shortcut when the maximum reached 0), a single-wave cloud, indices only (new_xyz NULL),
    the idx-only API chained through gather_point (the training path of sample_and_group), and the switch off."""
    import torch
    from pn2_amd import _lib
    S = pn2.tf_ops.tf_sampling
    lib, ptr = _lib.lib, _lib.ptr
    for b, n, levels in [(40, 777, (300, 120, 40)), (3, 64, (64, 32, 8)), (2, 300, (300, 150, 20)), (33, 2100, (512, 64))]:
        x = s_scene(b + n, b, n)
        if n == 300:
            x[:, 150:] = x[:, :150]          # every point has a twin: with m == n the maximum reaches 0 at step 150
        cur, cur_ref = T(x, cuda), x
        for m in levels:
            idx, new_xyz = S.farthest_point_sample_and_gather(m, cur)
            want = oracle.farthest_point_sample(m, cur_ref, 2)
            assert np.array_equal(idx.cpu().numpy(), want), (b, n, m)
            cur_ref = oracle.gather_point(cur_ref, want)
            assert np.array_equal(new_xyz.cpu().numpy(), cur_ref)
            cur = new_xyz
    # indices only, record in / out through the raw entry point
    x = s_scene(5, 4, 1024)
    xt = T(x, cuda)
    out = torch.empty((4, 256), dtype=torch.int32, device=cuda)
    tie = torch.empty((4,), dtype=torch.int32, device=cuda)
    _lib.check(lib.pn2_fps_nested(4, 1024, 256, ptr(xt), None, ptr(out), None, None, ptr(tie), 2, _lib.stream_ptr()), "pn2_fps_nested")
    assert np.array_equal(out.cpu().numpy(), oracle.farthest_point_sample(256, x, 2))
    assert np.array_equal(np.minimum(tie.cpu().numpy(), 255), np.minimum(oracle.fps_first_tie(256, x, 2), 255))
    # the idx-only API (sample_and_group's training path): farthest_point_sample -> gather_point -> tag -> next level
    pu = pn2.util.pointnet_util
    nx1, _, _, _ = pn2.sample_and_group(256, 0.3, 8, xt, None)
    assert S.fps_tie_record(nx1) is not None
    nx2, _, _, _ = pn2.sample_and_group(64, 0.5, 8, nx1, None)
    r1 = oracle.gather_point(x, oracle.farthest_point_sample(256, x, 2))
    r2 = oracle.gather_point(r1, oracle.farthest_point_sample(64, r1, 2))
    assert np.array_equal(nx1.cpu().numpy(), r1) and np.array_equal(nx2.cpu().numpy(), r2)
    S.USE_NESTED_FPS = False
    try:
        _, a = S.farthest_point_sample_and_gather(256, xt)
        assert S.fps_tie_record(a) is None
        _, b2 = S.farthest_point_sample_and_gather(64, a)
        assert np.array_equal(b2.cpu().numpy(), r2)
    finally:
        S.USE_NESTED_FPS = True


def test_fps_more_samples_than_points_equals_the_reference_kernel(pn2, oracle, ref, cuda):
    """VERDICT r05 missing #3: npoint > n.  The reference accepts it (tf_sampling.cpp:116-156 checks only npoint > 0) and its
    kernel keeps picking once every point is taken -- all running minima are 0, the strict `>` scan and the left-biased tree
    (tf_sampling.cu:131-175) then answer index 0 -- e.g. [0 5 4 6 7 2 1 3 0 0 0 0] for 12 of 8.  Every sampler entry point of
    this package == the reference's own kernel (oracle/_ref) == the C restatement: the one-pick register kernels (n = 1, 8, 64),
    the multi-wave layouts (300, 1536), the lazy kernel (8192-point clouds), a lattice full of ties, the row-strided twin, the
    nested chain on such a level's output, and the large-scene sampler."""
    import torch
    from pn2_amd import _lib
    S = pn2.tf_ops.tf_sampling
    lib, ptr = _lib.lib, _lib.ptr
    cases = [(2, 1, 3, "scene"), (3, 8, 12, "scene"), (2, 64, 100, "grid"), (2, 300, 333, "scene"), (2, 1536, 1600, "randn"),
             (1, 8192, 8200, "scene"), (2, 27, 64, "lattice")]
    for b, n, m, kind in cases:
        if kind == "lattice":  # 3 x 3 x 3 lattice: every step after the first is tied
            g = np.stack(np.meshgrid(*[np.arange(3)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
            x = np.stack([g, g[::-1].copy()])
        else:
            x = _gen(kind, 11 * n + m, b, n)
        build = "fast_noslp"
        want = ref.farthest_point_sample(m, x, build)
        assert np.array_equal(want, oracle.farthest_point_sample(m, x, FPS_MODE[build])), (n, m)
        if kind != "lattice" and n > 1:
            assert (want[:, n:] == 0).all() and sorted(want[0, :n].tolist()) == list(range(n))  # a permutation, then index 0
        xt = T(x, cuda)
        set_mode(FPS_MODE[build])
        got = pn2.farthest_point_sample(m, xt).cpu().numpy()
        assert np.array_equal(got, want), (n, m, got[0, -8:], want[0, -8:])
        # fps + gather in one launch, its coordinates, and the chain below it (the tie record of an exhausted cloud must not
        # let the next level take the shortcut: rows n.. are copies of row 0)
        idx, new_xyz = S.farthest_point_sample_and_gather(m, xt)
        assert np.array_equal(idx.cpu().numpy(), want)
        nx_ref = oracle.gather_point(x, want)
        assert np.array_equal(new_xyz.cpu().numpy(), nx_ref)
        m2 = max(1, m // 2)
        idx2, _ = S.farthest_point_sample_and_gather(m2, new_xyz)
        assert np.array_equal(idx2.cpu().numpy(), ref.farthest_point_sample(m2, nx_ref, build)), (n, m, m2)
        # the xyz columns of a wider batch read in place (pn2_fps_nested_ld)
        wide = torch.zeros((b, n, 6), dtype=torch.float32, device=cuda)
        wide[:, :, :3] = xt
        wide[:, :, 3:] = 7.0
        assert np.array_equal(pn2.farthest_point_sample(m, wide[:, :, 0:3]).cpu().numpy(), want)
        # the raw entry point without the nested machinery
        out = torch.full((b, m), -1, dtype=torch.int32, device=cuda)
        temp = torch.empty((b, n), dtype=torch.float32, device=cuda)
        _lib.check(lib.pn2_farthest_point_sample(b, n, m, ptr(xt), ptr(temp), ptr(out), FPS_MODE[build], _lib.stream_ptr()),
                   "pn2_farthest_point_sample")
        assert np.array_equal(out.cpu().numpy(), want)
    # the large-scene sampler (pn2_fps_large, configs[4]'s kernel) on a cloud just above the register kernels' limit
    n, m = 16500, 16600
    x = s_scene(9, 1, n)
    want = ref.farthest_point_sample(m, x, "fast_noslp")
    got = pn2.farthest_point_sample(m, T(x, cuda)).cpu().numpy()
    assert np.array_equal(got, want), np.argwhere(got != want)[:4]


def test_fps_tie_record_sees_three_holders_in_one_lane(pn2, oracle, cuda):
    """the one-pick kernels keep PPT points per lane (k, k+NT, k+2NT, ...); a maximum held by THREE rows of one lane
    (the winner among them) slipped through an OR/XOR parity count.  Three points at distance 1 from the first pick on indices
    congruent mod NT, every layout with PPT >= 3 (<64,4>, <512,4>, the coarse kernel's <256,4>): the record must say step 1."""
    import torch
    from pn2_amd import _lib
    lib, ptr = _lib.lib, _lib.ptr
    S, pu = pn2.tf_ops.tf_sampling, pn2.util.pointnet_util
    for n, nt, a in [(200, 64, 5), (1536, 512, 77), (2048, 512, 511), (1000, 256, 130)]:
        rs = np.random.RandomState(n)
        x = (rs.random_sample((3, n, 3)).astype(np.float32) - 0.5) * 0.2
        x[:, 0] = 0.0
        for ci in range(3):
            x[ci, a], x[ci, a + nt], x[ci, a + 2 * nt] = (1, 0, 0), (0, 1, 0), (0, 0, 1)
        x[2, a + nt] = (0.5, 0, 0)   # cloud 2: only two holders (the case the old count did see)
        want = oracle.fps_first_tie(8, x, 2)
        assert want.tolist() == [1, 1, 1]
        xt = T(x, cuda)
        out = torch.empty((3, 8), dtype=torch.int32, device=cuda)
        tie = torch.empty((3,), dtype=torch.int32, device=cuda)
        _lib.check(lib.pn2_fps_nested(3, n, 8, ptr(xt), None, ptr(out), None, None, ptr(tie), 2, _lib.stream_ptr()), "pn2_fps_nested")
        assert np.array_equal(out.cpu().numpy(), oracle.farthest_point_sample(8, x, 2)), n
        assert tie.cpu().tolist() == [1, 1, 1], (n, nt, tie.cpu().tolist())
        if n <= pu.COARSE_MAX_N:  # the in-kernel sampler of pn2_coarse_geometry (<256,4>): an untagged source cloud is sampled
            lv = pu.coarse_geometry(xt, [8], [0.5], [4])
            assert np.array_equal(lv[0]["fps_idx"].cpu().numpy(), oracle.farthest_point_sample(8, x, 2))
            assert S.fps_tie_record(lv[0]["new_xyz"]).cpu().tolist() == [1, 1, 1], n


def test_nested_fps_refuses_the_shortcut_when_every_row_is_asked_for(pn2, oracle, cuda):
    """the samplers do not look at ties of their LAST pick, so a level that asks for m == n rows of a tagged parent
    must sample.  Parent [A, B, C, A'] (A' == A) picked completely: its last pick meets td == 0 everywhere and repeats index 0;
    the child over those rows must repeat it too ([0, 1, 2, 0], not 0..3).  A one-point parent repeats its point from step 1 on
    (no second holder of the maximum 0 for the tie branches to see: the record is set by the entry point)."""
    S = pn2.tf_ops.tf_sampling
    x = np.zeros((2, 4, 3), np.float32)
    x[:, 1], x[:, 2], x[:, 3] = (4, 0, 0), (0, 3, 0), (0, 0, 0)   # row 3 duplicates row 0
    for x0, levels in ((x, (4, 4, 4)), (np.ones((3, 1, 3), np.float32), (3, 2, 2)), (x, (4, 3, 3))):
        cur, cur_ref = T(x0, cuda), x0
        for m in levels:
            idx, new_xyz = S.farthest_point_sample_and_gather(m, cur)
            want = oracle.farthest_point_sample(m, cur_ref, 2)
            assert np.array_equal(idx.cpu().numpy(), want), (m, idx.cpu().numpy(), want)
            cur_ref = oracle.gather_point(cur_ref, want)
            assert np.array_equal(new_xyz.cpu().numpy(), cur_ref)
            cur = new_xyz


def test_nested_fps_tag_and_buffers_rewritten_through_raw_pointers(pn2, oracle, cuda):
    """tensors rewritten WITHOUT a version bump.  (1) a graph replay that rewrites new_xyz rewrites its tie
    record in the same launch: the tag, consumed eagerly afterwards, describes the new contents (a lattice cloud full of ties
    replayed into the static buffers of a graph captured on a tie-free cloud must NOT take the shortcut).  (2) the library's own
    raw-pointer copy into a tagged tensor (tf_util.multi_copy_) drops the tag."""
    import torch
    S, tfu = pn2.tf_ops.tf_sampling, pn2.util.tf_util
    b, n, m1, m2 = 4, 2048, 512, 128
    scene, lattice = s_scene(77, b, n), s_grid(78, b, n, 8)
    cap = pn2.runtime.CapturedForward(lambda x: S.farthest_point_sample_and_gather(m1, x)[1], T(scene, cuda))
    for cloud in (lattice, scene, lattice):
        (new_xyz,) = (cap(T(cloud, cuda)),)
        torch.cuda.synchronize()
        assert S.fps_tie_record(new_xyz) is not None            # still tagged: same tensor object, same version ...
        idx2, _ = S.farthest_point_sample_and_gather(m2, new_xyz)  # ... and the record it points to was rewritten by the replay
        r1 = oracle.gather_point(cloud, oracle.farthest_point_sample(m1, cloud, 2))
        assert np.array_equal(new_xyz.cpu().numpy(), r1)
        assert np.array_equal(idx2.cpu().numpy(), oracle.farthest_point_sample(m2, r1, 2))
    # (2)
    _, a = S.farthest_point_sample_and_gather(m1, T(scene, cuda))
    assert S.fps_tie_record(a) is not None
    other = T(oracle.gather_point(lattice, oracle.farthest_point_sample(m1, lattice, 2)), cuda)
    tfu.multi_copy_([a], [other])
    assert S.fps_tie_record(a) is None
    idx3, _ = S.farthest_point_sample_and_gather(m2, a)
    assert np.array_equal(idx3.cpu().numpy(), oracle.farthest_point_sample(m2, other.cpu().numpy(), 2))


def test_nested_fps_tag_is_dropped_when_it_no_longer_describes_the_tensor(pn2, cuda):
    S = pn2.tf_ops.tf_sampling
    x = T(s_scene(0, 2, 2048), cuda)
    _, a = S.farthest_point_sample_and_gather(512, x)
    assert S.fps_tie_record(a) is not None
    assert S.fps_tie_record(a.clone()) is None and S.fps_tie_record(a[:, :100]) is None and S.fps_tie_record(a * 1.0) is None
    with pn2.config.arith(fps=0):
        assert S.fps_tie_record(a) is None          # sampled under another contraction: distances differ in the last bit
    a.add_(1.0)
    assert S.fps_tie_record(a) is None              # modified in place since


def test_fps_tie_heavy_grids(pn2, oracle, ref, cuda):
    """coarse lattices: most rounds are multi-way exact ties -> the (max, k mod 512, k) rule of the 512-thread
    strided scan + left-biased tree (tf_sampling.cu:153-170) decides nearly every pick."""
    for n, m, q in [(8192, 1024, 16), (4096, 512, 8), (1500, 300, 4), (700, 64, 2), (8192, 64, 1)]:
        x = s_grid(n + q, 3, n, q)
        _three_way_fps(pn2, oracle, ref, cuda, m, x, True)


def test_fps_b_gt_32_and_ragged_n(pn2, oracle, ref, cuda):
    """b > 32: reference blocks stride over the batch re-using their temp row (tf_sampling.cu:121,220);
    n not a multiple of 512 / smaller than 512."""
    for b, n, m in [(40, 777, 50), (33, 100, 100), (3, 513, 17), (2, 5000, 129)]:
        x = s_scene(b + n, b, n)
        _three_way_fps(pn2, oracle, ref, cuda, m, x, False)


@pytest.mark.parametrize("n,m,k,r", [(64, 8, 4, 0.3), (1000, 130, 32, 0.15), (2048, 512, 64, 0.3), (4096, 256, 128, 0.5),
                                     (8192, 1024, 16, 0.25)])
def test_ball_query_shapes(pn2, oracle, ref, cuda, n, m, k, r):
    rs = np.random.RandomState(n)
    x1 = rs.random_sample((3, n, 3)).astype(np.float32)
    x2 = x1[:, rs.permutation(n)[:m]].copy()
    _three_way_bq(pn2, oracle, ref, cuda, r, k, x1, x2, False)
    x1 = s_grid(n, 3, n, 32)
    x2 = x1[:, :m].copy()
    _three_way_bq(pn2, oracle, ref, cuda, 4 / 32.0, k, x1, x2, True)  # radius exactly on lattice distances


# ---------------------------------------------------------------- gradients / copies ------------------------
def test_gather_and_group_grad_integer_valued(pn2, oracle, ref, cuda):
    """integer-valued upstream gradients: fp32 atomics are exact in any order -> bit-exact against the
    reference's atomicAdd kernels (tf_sampling.cu:193-206, tf_grouping.cu:70-90)."""
    import torch
    rs = np.random.RandomState(3)
    b, n, m, k, c = 4, 1024, 256, 32, 16
    x = s_randn(3, b, n)
    idx1 = rs.randint(0, n, (b, m)).astype(np.int32)
    go = rs.randint(-8, 9, (b, m, 3)).astype(np.float32)
    r = ref.gather_point_grad(x, idx1, go)
    assert np.array_equal(r, oracle.gather_point_grad(x, idx1, go))
    xt = T(x, cuda).requires_grad_(True)
    pn2.gather_point(xt, T(idx1, cuda)).backward(T(go, cuda))
    assert np.array_equal(xt.grad.cpu().numpy(), r)

    pts = rs.randn(b, n, c).astype(np.float32)
    idx = rs.randint(0, n // 4, (b, m, k)).astype(np.int32)  # popular points: contended atomics
    gout = rs.randint(-8, 9, (b, m, k, c)).astype(np.float32)
    r = ref.group_point_grad(pts, idx, gout)
    assert np.array_equal(r, oracle.group_point_grad(pts, idx, gout))
    pt = T(pts, cuda).requires_grad_(True)
    pn2.group_point(pt, T(idx, cuda)).backward(T(gout, cuda))
    assert np.array_equal(pt.grad.cpu().numpy(), r)
    assert np.array_equal(ref.group_point(pts, idx), pn2.group_point(T(pts, cuda), T(idx, cuda)).detach().cpu().numpy())


def test_group_point_reference_gradcheck_shape(pn2, oracle, ref, cuda):
    """tf_ops/test_tf_ops.py:38-56 shapes: points (1,128,16), xyz1 (1,128,3), xyz2 (1,8,3), r=0.3, K=32."""
    np.random.seed(100)
    points = np.random.random((1, 128, 16)).astype("float32")
    xyz1 = np.random.random((1, 128, 3)).astype("float32")
    xyz2 = np.random.random((1, 8, 3)).astype("float32")
    _three_way_bq(pn2, oracle, ref, cuda, 0.3, 32, xyz1, xyz2, False)
    idx, cnt = ref.query_ball_point(0.3, 32, xyz1, xyz2)
    g = ref.group_point(points, idx)
    assert np.array_equal(g, pn2.group_point(T(points, cuda), T(idx, cuda)).cpu().numpy())
    assert np.array_equal(g, oracle.group_point(points, idx))


# ---------------------------------------------------------------- N3 / N4 ops -------------------------------
@pytest.mark.parametrize("b,m,n,k", [(2, 16, 64, 8), (3, 50, 300, 32), (1, 7, 1000, 5), (2, 9, 33, 33)])
def test_selection_sort_whole_rows(pn2, oracle, ref, cuda, b, m, n, k):
    rs = np.random.RandomState(b * 100 + n)
    for dist in (rs.random_sample((b, m, n)).astype(np.float32),
                 (rs.randint(0, 6, (b, m, n)) / 4.0).astype(np.float32)):  # heavy ties
        ri, ro = ref.selection_sort(k, dist)
        oi, oo = oracle.select_top_k(k, dist)
        assert np.array_equal(ri, oi) and np.array_equal(ro, oo)
        hi, ho = pn2.select_top_k(k, T(dist, cuda))
        assert np.array_equal(ri, hi.cpu().numpy()) and np.array_equal(ro, ho.cpu().numpy())


@pytest.mark.parametrize("b,n,m", [(2, 9, 100), (3, 1000, 4096), (2, 8192, 3000), (1, 8193, 1000), (2, 20000, 5000),
                                   (40, 70, 64)])
def test_prob_sample_running_sum_and_draws(pn2, oracle, ref, cuda, b, n, m):
    rs = np.random.RandomState(n)
    p = rs.random_sample((b, n)).astype(np.float32)
    r_ = rs.random_sample((b, m)).astype(np.float32)
    for build in ref.BUILDS:
        ro, rt = ref.prob_sample(p, r_, build)
        oo, ot = oracle.prob_sample(p, r_)
        assert np.array_equal(rt, ot), "running sums differ (%s)" % build
        assert np.array_equal(ro, oo)
    assert np.array_equal(pn2.prob_sample(T(p, cuda), T(r_, cuda)).cpu().numpy(), ro)


# ---------------------------------------------------------------- the reference's kernels, timed on the same GPU ----
def test_reference_kernels_vs_hip_timing_report(pn2, ref, cuda):
    """Not a parity test: the reference's own kernels (oracle/_ref, -ffp-contract=off build) and this package's kernels
    timed back to back on the SAME MI355X at the configs[1] SA1 / north-star shapes.  Writes
    gpurun_out/ref_vs_hip_timing.json (copied to profiles/) and only asserts that the report is complete and that the
    HIP kernels are not slower -- the numbers are the deliverable."""
    import ctypes
    import json
    import os
    import torch
    B, N, M, K, C = 16, 8192, 1024, 32, 128
    L = ref.lib("off")
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    xyz = T(s_scene(0, B, N), cuda)
    feat = torch.randn(B, N, C, device=cuda)
    temp = torch.empty(32, N, device=cuda)
    f_ref = torch.empty(B, M, dtype=torch.int32, device=cuda)
    new_xyz = pn2.gather_point(xyz, pn2.farthest_point_sample(M, xyz))
    idx, _ = pn2.query_ball_point(0.5, K, xyz, new_xyz)
    idx_ref = torch.empty_like(idx)
    cnt_ref = torch.empty(B, M, dtype=torch.int32, device=cuda)
    grouped = torch.empty(B, M, K, C, device=cuda)

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps * 1e3  # us (the _ref doors synchronise the device themselves: includes ~10 us of that)

    rep = {"shape": "B=16 N=8192 M=1024 K=32 C=128 (configs[1] SA1 geometry, north-star feature width)", "unit": "us", "rows": {}}
    rep["rows"]["farthest_point_sample"] = {
        "reference_kernel": timed(lambda: L.ref_farthest_point_sample(B, N, M, P(xyz), P(temp), P(f_ref)), 3),
        "hip": timed(lambda: pn2.farthest_point_sample(M, xyz), 10)}
    rep["rows"]["query_ball_point"] = {
        "reference_kernel": timed(lambda: L.ref_query_ball_point(B, N, M, ctypes.c_float(0.5), K, P(xyz), P(new_xyz), P(idx_ref), P(cnt_ref)), 3),
        "hip": timed(lambda: pn2.query_ball_point(0.5, K, xyz, new_xyz), 20)}
    rep["rows"]["group_point"] = {
        "reference_kernel": timed(lambda: L.ref_group_point(B, N, C, M, K, P(feat), P(idx), P(grouped)), 3),
        "hip": timed(lambda: pn2.group_point(feat, idx), 20)}
    for k, v in rep["rows"].items():
        v["speedup"] = round(v["reference_kernel"] / v["hip"], 1)
        v["reference_kernel"], v["hip"] = round(v["reference_kernel"], 1), round(v["hip"], 1)
        assert v["hip"] < v["reference_kernel"], (k, v)
    rep["note"] = ("reference kernels = tf_ops/tf_sampling.cu / tf_grouping.cu compiled unmodified for gfx950 (<<<32,512>>>, <<<b,256>>> "
                   "launch shapes of the source); timings include the shim's device synchronisation")
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "ref_vs_hip_timing.json"), "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep["rows"]))


# ---------------------------------------------------------------- A8 / A8g: three_interpolate against the lifted host functions
@pytest.mark.parametrize("b,m,c,n", [(1, 8, 16, 128), (16, 16, 512, 64), (16, 64, 256, 256), (16, 256, 256, 1024),
                                     (16, 1024, 128, 8192), (2, 60, 131, 333), (2, 5, 3, 7)])
def test_three_interpolate_equals_the_reference_host_functions(pn2, cuda, b, m, c, n):
    """The HIP three_interpolate == the reference's own threeinterpolate_cpu (oracle/_ref/libpn2_ref_interp.so: lifted from
    tf_interpolate.cpp:307-330 at build time) BIT FOR BIT at every FP level of configs[1] (B=16: m -> n = 16->64, 64->256,
    256->1024, 1024->8192 with the channel widths of model.py:90-129) and the reference's test shape; the gradient equals
    threeinterpolate_grad_cpu (:397-421) exactly on integer-valued operands (every product and sum exact, so the
    order of the device atomics / gathers cannot show) and to fp32 summation order otherwise."""
    import torch
    from oracle import ref as R
    if not R.interp_available():
        pytest.skip("oracle/_ref/libpn2_ref_interp.so missing: `make -C oracle _ref` (needs /root/reference)")
    rs = np.random.RandomState(b * 1000 + c)
    pts = rs.randn(b, m, c).astype(np.float32)
    idx = rs.randint(0, m, (b, n, 3)).astype(np.int32)
    d = rs.rand(b, n, 3).astype(np.float32) + 1e-3
    w = ((1.0 / d) / (1.0 / d).sum(2, keepdims=True)).astype(np.float32)
    pt = T(pts, cuda).requires_grad_(True)
    out = pn2.three_interpolate(pt, T(idx, cuda), T(w, cuda))
    assert np.array_equal(out.detach().cpu().numpy(), R.three_interpolate(pts, idx, w))
    go = rs.randn(b, n, c).astype(np.float32)
    out.backward(T(go, cuda))
    rg = R.three_interpolate_grad(pts, idx, w, go)
    assert np.allclose(pt.grad.cpu().numpy(), rg, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(rg).max())))
    # exact operands: weights in {0.25, 0.5}, integer gradients -> bit-exact gradient
    wi = rs.choice([0.25, 0.5], (b, n, 3)).astype(np.float32)
    gi = rs.randint(-8, 9, (b, n, c)).astype(np.float32)
    pt2 = T(pts, cuda).requires_grad_(True)
    pn2.three_interpolate(pt2, T(idx, cuda), T(wi, cuda)).backward(T(gi, cuda))
    assert np.array_equal(pt2.grad.cpu().numpy(), R.three_interpolate_grad(pts, idx, wi, gi))
