"""GPU parity tests: every HIP op, called through the C ABI (ctypes), against the CPU oracle on
the same seeded inputs.  Index outputs must be bit-exact; float copies bit-exact; scatter-add
gradients within fp32 summation-order tolerance."""
import os

import numpy as np
import pytest

from conftest import s_grid, s_randn, s_scene

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def T(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


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


# ------------------------------------------------------------------ FPS -------------------
@pytest.mark.parametrize("n,m", [(64, 16), (256, 64), (500, 100), (1024, 256), (1500, 200), (3000, 300),
                                 (4096, 512), (8192, 1024), (10000, 64)])
def test_fps_grid_bit_exact_all_modes(pn2, oracle, cuda, n, m):
    x = s_grid(n, 3, n, 64 if n < 5000 else 1024)  # coarse grid: exact arithmetic + many ties
    ref = oracle.farthest_point_sample(m, x, 0)
    for mode in (0, 1, 2):
        set_mode(mode)
        got = pn2.farthest_point_sample(m, T(x, cuda)).cpu().numpy()
        assert got.dtype == np.int32 and got.shape == (3, m)
        assert np.array_equal(got, ref), "mode %d first diff at %s" % (mode, np.argwhere(got != ref)[:3])


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("gen", ["randn", "scene"])
def test_fps_float_inputs_bit_exact_per_mode(pn2, oracle, cuda, mode, gen):
    x = s_randn(11, 4, 2048) if gen == "randn" else s_scene(12, 4, 2048)
    set_mode(mode)
    got = pn2.farthest_point_sample(512, T(x, cuda)).cpu().numpy()
    assert np.array_equal(got, oracle.farthest_point_sample(512, x, mode))


def test_fps_duplicates_and_degenerate(pn2, oracle, cuda):
    x = np.ones((2, 700, 3), np.float32)  # all identical: every round is an all-way tie
    assert pn2.farthest_point_sample(6, T(x, cuda)).cpu().numpy().tolist() == [[0] * 6] * 2
    x = s_grid(5, 2, 1300, 16)  # duplicated points (dataset up-sampling duplicates short clouds)
    assert np.array_equal(pn2.farthest_point_sample(300, T(x, cuda)).cpu().numpy(),
                          oracle.farthest_point_sample(300, x))
    x = s_randn(6, 1, 3)  # n < one wave, m == n
    assert np.array_equal(pn2.farthest_point_sample(3, T(x, cuda)).cpu().numpy(),
                          oracle.farthest_point_sample(3, x))
    assert pn2.farthest_point_sample(1, T(x, cuda)).cpu().numpy().tolist() == [[0]]


@pytest.mark.parametrize("case", ["ragged", "m_eq_n", "all_equal", "two_values", "coarse_lattice", "clusters", "line",
                                  "far_outlier", "m_small"])
def test_fps_lazy_multipick_kernel_paths(pn2, oracle, cuda, case):
    """The lazy multi-pick kernel (2048 < n <= 8192: candidate list above a threshold, picks without a pass over the
    cloud, pending picks applied through bounding-box tests) on the inputs that drive its special paths: list overflow
    and empty lists (lattices, duplicates: the single-pick fallback), tau = 0 (all distances zero), clouds that are not a
    multiple of the 64-point buckets, m == n (list capacity vs pick count), degenerate boxes (a line), a far outlier
    (cell grid collapses), m below one phase.  Bit-exact against the oracle in all three arithmetic modes."""
    rs = np.random.RandomState(31)
    if case == "ragged":
        x, m = s_scene(31, 3, 5003), 700
    elif case == "m_eq_n":
        x, m = s_randn(32, 2, 2100), 2100
    elif case == "all_equal":
        x, m = np.full((2, 3000, 3), 0.25, np.float32), 40
    elif case == "two_values":
        x, m = (rs.randint(0, 2, (2, 4096, 3))).astype(np.float32), 64
    elif case == "coarse_lattice":
        x, m = s_grid(33, 2, 8192, 8), 600  # 512 distinct positions, 16 copies each: ties everywhere
    elif case == "clusters":
        c = rs.uniform(-20, 20, (2, 12, 1, 3))
        x, m = (c + 0.05 * rs.randn(2, 12, 512, 3)).reshape(2, 6144, 3).astype(np.float32), 512
    elif case == "line":
        x = np.zeros((2, 4100, 3), np.float32)
        x[:, :, 1] = rs.uniform(-3, 3, (2, 4100))
        m = 300
    elif case == "far_outlier":
        x = s_scene(34, 2, 8192)
        x[:, 77] = [4.0e5, -3.0e5, 1.0e5]
        m = 256
    else:
        x, m = s_scene(35, 4, 8192), 2
    ref = {}
    for mode in (0, 1, 2):
        set_mode(mode)
        ref = oracle.farthest_point_sample(m, x, mode)
        idx, nx = pn2.tf_ops.tf_sampling.farthest_point_sample_and_gather(m, T(x, cuda))
        got = idx.cpu().numpy()
        assert np.array_equal(got, ref), "%s mode %d first diff at %s" % (case, mode, np.argwhere(got != ref)[:3])
        assert np.array_equal(nx.cpu().numpy(), np.take_along_axis(x, ref[:, :, None].astype(np.int64), 1))


def test_fps_random_shapes_differential_fuzz(pn2, oracle, cuda):
    """Random (b, n, m, distribution) draws across every kernel boundary of the dispatcher -- one-pick register kernels
    (n <= 2048), lazy multi-pick (<= 8192), streaming (<= 16384), Hilbert-bucket lazy kernel (pn2_fps_large) -- against the
    oracle (tf_sampling.cu:111-176), default arithmetic mode; also pn2_fps_gather's new_xyz."""
    rs = np.random.RandomState(77)
    sizes = [2049, 2111, 3000, 4097, 6007, 8191, 8193, 12000, 16384, 16385, 20011, 40000]
    for n in sizes:
        b = int(rs.randint(1, 4))
        m = int(rs.choice([1, 2, 63, 65, 300, 777, min(n, 1500)]))
        kind = rs.randint(0, 4)
        if kind == 0:
            x = s_scene(n, b, n)
        elif kind == 1:
            x = s_randn(n, b, n) * np.float32(rs.choice([1e-3, 1.0, 1e3]))
        elif kind == 2:
            x = (rs.randint(-6, 7, (b, n, 3)) * 0.5).astype(np.float32)          # lattice: ties, duplicates
        else:
            x = np.concatenate([s_randn(n, b, n - n // 3), np.repeat(s_randn(n + 1, b, 1), n // 3, 1) +
                                1e-4 * s_randn(n + 2, b, n // 3)], 1).astype(np.float32)  # a third of the cloud in one tight clump
        ref = oracle.farthest_point_sample(m, x)
        xt = T(x, cuda)
        idx, nx = pn2.tf_ops.tf_sampling.farthest_point_sample_and_gather(m, xt)  # n > 16384 -> pn2_fps_large
        got = idx.cpu().numpy()
        assert np.array_equal(got, ref), "n=%d m=%d kind=%d first diff at %s" % (n, m, kind, np.argwhere(got != ref)[:3])
        assert np.array_equal(nx.cpu().numpy(), np.take_along_axis(x, ref[:, :, None].astype(np.int64), 1))


def test_fps_streaming_kernel_large_n(pn2, oracle, cuda):
    x = s_scene(8, 2, 20000)  # n > 16384 -> global-scratch kernel
    assert np.array_equal(pn2.farthest_point_sample(40, T(x, cuda)).cpu().numpy(),
                          oracle.farthest_point_sample(40, x))
    x = s_grid(9, 34, 17000, 32)[:, :, :]  # b > 32: blocks stride over the batch with the (32,n) scratch
    assert np.array_equal(pn2.farthest_point_sample(6, T(x, cuda)).cpu().numpy(),
                          oracle.farthest_point_sample(6, x))


def test_fps_full_size_sa1(pn2, oracle, cuda):
    """BASELINE config[1] SA1 shape B=16,N=8192,M=1024: all 16 scenes against the oracle + properties."""
    import torch
    x = s_scene(0, 16, 8192)
    got = pn2.farthest_point_sample(1024, T(x, cuda)).cpu().numpy()
    ref = oracle.farthest_point_sample(1024, x)
    assert np.array_equal(got, ref)
    assert (got[:, 0] == 0).all()
    for b in range(16):
        assert len(np.unique(got[b])) == 1024  # continuous data: no duplicates
    # farthest-point property: the distance of each pick to the already-picked set is non-increasing
    p = x[3][got[3]].astype(np.float64)
    d = np.full(1024, np.inf)
    seq = []
    for j in range(1, 200):
        d = np.minimum(d, ((p - p[j - 1]) ** 2).sum(1))
        seq.append(d[j])
    assert all(seq[i] >= seq[i + 1] - 1e-9 for i in range(len(seq) - 1))


def test_golden_fixtures_on_gpu(pn2, cuda):
    g = np.load(os.path.join(GOLD, "oracle_small.npz"))
    rs = np.random.RandomState(7)
    xyz = rs.random_sample((2, 1024, 3)).astype(np.float32)
    for mode in (0, 1, 2):
        set_mode(mode)
        f = pn2.farthest_point_sample(256, T(xyz, cuda))
        assert np.array_equal(f.cpu().numpy(), g["cfg0_fps_m%d" % mode])
        idx, cnt = pn2.query_ball_point(0.2, 16, T(xyz, cuda), pn2.gather_point(T(xyz, cuda), f))
        assert np.array_equal(idx.cpu().numpy(), g["cfg0_bq_idx_m%d" % mode])
        assert np.array_equal(cnt.cpu().numpy(), g["cfg0_bq_cnt_m%d" % mode])


def test_three_nn_reference_golden_vector_on_gpu(pn2, cuda):
    """The reference's own KAT (tf_ops/test_interpolate.py:30-35) through the HIP kernel."""
    g = np.load(os.path.join(GOLD, "reference_three_nn.npz"))
    np.random.seed(int(g["seed"]))
    target = np.random.random(tuple(g["target_shape"])).astype("float32")
    reference = np.random.random(tuple(g["reference_shape"])).astype("float32")
    dist, idx = pn2.three_nn(T(target[:4], cuda), T(reference[:4], cuda))
    dist, idx = dist.cpu().numpy(), idx.cpu().numpy()
    assert (idx[:3, :3, :1].flatten() == g["idx"]).all()
    assert np.array2string(dist[:3, :3, :1].flatten()) == str(g["printed"])


# ------------------------------------------------------------------ gather ----------------
def test_gather_point_and_grad(pn2, oracle, cuda):
    import torch
    rs = np.random.RandomState(1)
    x = s_randn(1, 3, 777)
    idx = rs.randint(0, 777, (3, 130)).astype(np.int32)
    xt = T(x, cuda).requires_grad_(True)
    out = pn2.gather_point(xt, T(idx, cuda))
    assert np.array_equal(out.detach().cpu().numpy(), oracle.gather_point(x, idx))
    go = rs.randn(3, 130, 3).astype(np.float32)
    out.backward(T(go, cuda))
    assert np.allclose(xt.grad.cpu().numpy(), oracle.gather_point_grad(x, idx, go), rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ ball query ------------
@pytest.mark.parametrize("n,m,ns,r", [(64, 8, 4, 0.3), (300, 37, 16, 0.2), (1000, 130, 32, 0.15), (1500, 200, 32, 0.25),
                                      (2048, 256, 64, 0.3), (130, 130, 8, 2.0)])
def test_ball_query_grid_bit_exact_all_modes(pn2, oracle, cuda, n, m, ns, r):
    x = s_grid(n + 1, 2, n, 64)
    q = x[:, :m].copy()
    q[:, ::3] += np.float32(1.0 / 128)  # off-grid-by-half queries too (still exact arithmetic)
    ri, rc = oracle.query_ball_point(r, ns, x, q, 0)
    for mode in (0, 1, 2):
        set_mode(mode)
        gi, gc = pn2.query_ball_point(r, ns, T(x, cuda), T(q, cuda))
        assert np.array_equal(gc.cpu().numpy(), rc)
        assert np.array_equal(gi.cpu().numpy(), ri)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_ball_query_float_inputs(pn2, oracle, cuda, mode):
    set_mode(mode)
    for x, r in ((s_randn(21, 3, 3000), 0.5), (s_scene(22, 3, 3000), 0.5), (s_scene(23, 3, 3000), 2.0)):
        f = oracle.farthest_point_sample(200, x, mode)
        q = oracle.gather_point(x, f)
        gi, gc = pn2.query_ball_point(r, 32, T(x, cuda), T(q, cuda))
        ri, rc = oracle.query_ball_point(r, 32, x, q, mode)
        assert np.array_equal(gc.cpu().numpy(), rc) and np.array_equal(gi.cpu().numpy(), ri)
        assert rc.min() >= 1  # queries are dataset points


def test_ball_query_radius_boundary_is_exact(pn2, oracle, cuda):
    """Distances exactly equal to / one ulp around the radius: the sqrt-free threshold must agree
    with `sqrtf(d2) < r` (strict) for every candidate."""
    rs = np.random.RandomState(5)
    for r in (0.25, 0.3, 0.1, 1.0, 0.7071068, 3.0):
        r32 = np.float32(r)
        # points on the x axis at distances r-2ulp .. r+2ulp from the query at the origin
        ds = [r32]
        for _ in range(3):
            ds.append(np.nextafter(ds[-1], np.float32(10)))
        lo = r32
        for _ in range(3):
            lo = np.nextafter(lo, np.float32(0))
            ds.append(lo)
        pts = np.zeros((1, 64, 3), np.float32)
        pts[0, :len(ds), 0] = np.array(ds, np.float32)
        pts[0, len(ds):, :] = 50 + rs.rand(64 - len(ds), 3)
        q = np.zeros((1, 1, 3), np.float32)
        pts[0, 20] = 0  # the query itself
        for mode in (0, 1, 2):
            set_mode(mode)
            gi, gc = pn2.query_ball_point(float(r32), 16, T(pts, cuda), T(q, cuda))
            ri, rc = oracle.query_ball_point(float(r32), 16, pts, q, mode)
            assert np.array_equal(gc.cpu().numpy(), rc) and np.array_equal(gi.cpu().numpy(), ri), (r, mode)


def test_ball_query_empty_ball_and_full_ball(pn2, oracle, cuda):
    x = s_grid(31, 1, 500, 64)
    q = np.concatenate([np.full((1, 1, 3), 100.0, np.float32), x[:, :3]], axis=1)
    gi, gc = pn2.query_ball_point(0.1, 8, T(x, cuda), T(q, cuda))
    gi, gc = gi.cpu().numpy(), gc.cpu().numpy()
    assert gc[0, 0] == 0 and (gi[0, 0] == 0).all()  # documented: empty rows are zero-filled
    ri, rc = oracle.query_ball_point(0.1, 8, x, q)
    assert np.array_equal(gi, ri) and np.array_equal(gc, rc)
    gi, gc = pn2.query_ball_point(10.0, 8, T(x, cuda), T(q[:, 1:], cuda))  # everything is inside: first 8 indices
    assert (gi.cpu().numpy() == np.arange(8)).all() and (gc.cpu().numpy() == 8).all()


def test_ball_query_full_size_sa1(pn2, oracle, cuda):
    x = s_scene(0, 16, 8192)
    f = oracle.farthest_point_sample(1024, x[:2])
    fx = pn2.farthest_point_sample(1024, T(x, cuda))
    q = pn2.gather_point(T(x, cuda), fx)
    gi, gc = pn2.query_ball_point(0.5, 32, T(x, cuda), q)
    gi, gc = gi.cpu().numpy(), gc.cpu().numpy()
    ri, rc = oracle.query_ball_point(0.5, 32, x[:2], oracle.gather_point(x[:2], f))
    assert np.array_equal(gi[:2], ri) and np.array_equal(gc[:2], rc)
    # size-independent properties on all 16 batch elements
    qn = q.cpu().numpy()
    assert gc.min() >= 1 and gc.max() <= 32
    for b in (5, 11, 15):
        for j in range(0, 1024, 97):
            c = gc[b, j]
            row = gi[b, j]
            assert (np.diff(row[:c]) > 0).all()          # index order, no repeats among real hits
            assert (row[c:] == row[0]).all()             # padding = first hit
            d = np.sqrt(((x[b][row[:c]] - qn[b, j]) ** 2).sum(1))
            assert (d < 0.5 + 1e-6).all()


# ------------------------------------------------------------------ group ------------------
@pytest.mark.parametrize("c", [1, 3, 4, 6, 64, 67, 128])
def test_group_point_bit_exact_and_grad(pn2, oracle, cuda, c):
    import torch
    rs = np.random.RandomState(c)
    pts = rs.randn(3, 500, c).astype(np.float32)
    idx = rs.randint(0, 500, (3, 70, 16)).astype(np.int32)
    pt = T(pts, cuda).requires_grad_(True)
    out = pn2.group_point(pt, T(idx, cuda))
    assert np.array_equal(out.detach().cpu().numpy(), oracle.group_point(pts, idx))
    go = rs.randn(*out.shape).astype(np.float32)
    out.backward(T(go, cuda))
    ref = oracle.group_point_grad(pts, idx, go)
    assert np.allclose(pt.grad.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


def test_group_point_reference_gradient_check_shape(pn2, oracle, cuda):
    """tf_ops/test_tf_ops.py:38-56: points (1,128,16), ball query r=0.3 K=32 on (1,128,3)/(1,8,3);
    the Jacobian of group_point w.r.t. points must match finite differences (< 1e-4).  group_point is
    linear in points, so J^T g == group_point_grad(g) exactly up to fp32 summation."""
    import torch
    rs = np.random.RandomState(0)
    points = rs.random_sample((1, 128, 16)).astype(np.float32)
    xyz1 = rs.random_sample((1, 128, 3)).astype(np.float32)
    xyz2 = rs.random_sample((1, 8, 3)).astype(np.float32)
    idx, cnt = pn2.query_ball_point(0.3, 32, T(xyz1, cuda), T(xyz2, cuda))
    ri, rc = oracle.query_ball_point(0.3, 32, xyz1, xyz2)
    assert np.array_equal(idx.cpu().numpy(), ri)
    pt = T(points, cuda).requires_grad_(True)
    out = pn2.group_point(pt, idx)
    # finite-difference check along random directions (linear op: exact up to rounding)
    for s in range(4):
        v = np.random.RandomState(s).randn(*points.shape).astype(np.float32)
        g = np.random.RandomState(10 + s).randn(*out.shape).astype(np.float32)
        (gp,) = torch.autograd.grad(out, pt, T(g, cuda), retain_graph=True)
        lhs = float((gp.cpu().numpy().astype(np.float64) * v).sum())
        rhs = float((oracle.group_point(v, ri).astype(np.float64) * g).sum())
        assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(rhs))


# ------------------------------------------------------------------ three_nn / interpolate --
@pytest.mark.parametrize("n,m", [(64, 16), (100, 3), (1000, 250), (3000, 2100), (8192, 1024)])
def test_three_nn_bit_exact(pn2, oracle, cuda, n, m):
    a = s_randn(n, 2, n)
    r = s_randn(m + 7, 2, m)
    d, i = pn2.three_nn(T(a, cuda), T(r, cuda))
    rd, ri = oracle.three_nn(a, r)
    assert np.array_equal(i.cpu().numpy(), ri)
    assert np.array_equal(d.cpu().numpy(), rd)


def test_three_nn_ties_lowest_index(pn2, oracle, cuda):
    a = s_grid(1, 2, 400, 8)
    r = s_grid(2, 2, 90, 8)  # 512 positions, many equidistant neighbours
    d, i = pn2.three_nn(T(a, cuda), T(r, cuda))
    rd, ri = oracle.three_nn(a, r)
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(d.cpu().numpy(), rd)


@pytest.mark.parametrize("case", ["offset_1000", "offset_1e5", "range_1e4", "duplicates", "scene", "collinear", "tiny_spacing"])
def test_three_nn_filter_adversarial(pn2, oracle, cuda, case):
    """The fp32 ranking filter of three_nn (expanded form around a centre, margin 64 u R^2) must never drop
    a true neighbour: far-from-origin clouds, huge extent/spacing ratios (margin useless -> the list
    overflows -> float64 fallback scan), exact duplicates (ties -> lowest index), degenerate geometry."""
    rs = np.random.RandomState(len(case))
    n, m = 700, 300
    if case == "offset_1000":
        r = rs.rand(2, m, 3).astype(np.float32) + np.float32(1000.0)
        a = rs.rand(2, n, 3).astype(np.float32) + np.float32(1000.0)
    elif case == "offset_1e5":  # spacing close to the fp32 grid at 1e5: massive ties
        r = (rs.rand(2, m, 3) * 4 + 1e5).astype(np.float32)
        a = (rs.rand(2, n, 3) * 4 + 1e5).astype(np.float32)
    elif case == "range_1e4":  # a tight cluster plus far outliers: R^2 u >> cluster distances
        r = (rs.rand(2, m, 3) * 1e-3).astype(np.float32)
        r[:, ::50] += np.float32(3000.0)
        a = (rs.rand(2, n, 3) * 1e-3).astype(np.float32)
        a[:, ::70] -= np.float32(2000.0)
    elif case == "duplicates":
        base = rs.rand(2, 10, 3).astype(np.float32)
        r = base[:, rs.randint(0, 10, m)]  # every known point repeated ~30 times
        a = base[:, rs.randint(0, 10, n)] + (rs.rand(2, n, 3) < 0.5).astype(np.float32) * np.float32(0.25)
    elif case == "scene":
        a = s_scene(3, 2, n)[..., :3]
        r = a[:, :m].copy()
    elif case == "collinear":
        t = rs.rand(2, m, 1).astype(np.float32)
        r = np.concatenate([t, t * np.float32(2), t * np.float32(-1)], 2)
        t = rs.rand(2, n, 1).astype(np.float32)
        a = np.concatenate([t, t * np.float32(2), t * np.float32(-1)], 2)
    else:  # tiny_spacing: denormal-scale differences around a large value
        r = (np.float32(8.0) + rs.randint(0, 64, (2, m, 3)).astype(np.float32) * np.float32(2.0 ** -20)).astype(np.float32)
        a = (np.float32(8.0) + rs.randint(0, 64, (2, n, 3)).astype(np.float32) * np.float32(2.0 ** -20)).astype(np.float32)
    a, r = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(r, np.float32)
    d, i = pn2.three_nn(T(a, cuda), T(r, cuda))
    rd, ri = oracle.three_nn(a, r)
    assert np.array_equal(i.cpu().numpy(), ri)
    assert np.array_equal(d.cpu().numpy(), rd)


@pytest.mark.parametrize("c", [1, 5, 16, 64, 128, 131])
def test_three_interpolate_bit_exact_and_grad(pn2, oracle, cuda, c):
    import torch
    rs = np.random.RandomState(c)
    pts = rs.randn(2, 60, c).astype(np.float32)
    idx = rs.randint(0, 60, (2, 333, 3)).astype(np.int32)
    w = rs.rand(2, 333, 3).astype(np.float32)
    w /= w.sum(2, keepdims=True)
    pt = T(pts, cuda).requires_grad_(True)
    out = pn2.three_interpolate(pt, T(idx, cuda), T(w, cuda))
    assert np.array_equal(out.detach().cpu().numpy(), oracle.three_interpolate(pts, idx, w))
    go = rs.randn(*out.shape).astype(np.float32)
    out.backward(T(go, cuda))
    ref = oracle.three_interpolate_grad(pts, idx, w, go)
    assert np.allclose(pt.grad.cpu().numpy(), ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("c,n", [(32, 2048), (128, 700), (64, 5), (260, 333)])
def test_group_point_grad_gather_path(pn2, oracle, cuda, c, n):
    """Levels large enough for the list-and-gather gradient (pn2_group_point_grad_ws): points nobody groups (zero
    rows), a hot point that pads many balls, every neighbour of a ball the same point."""
    import torch
    b, m, ns = 4, 512, 16
    rs = np.random.RandomState(c + n)
    idx = rs.randint(0, n, (b, m, ns)).astype(np.int32)
    idx[idx == 3 % n] = 0                      # point 3 never grouped (when n > 3)
    idx[:, ::5, ns // 2:] = 1 % n              # hot point: pads the second half of every fifth ball
    idx[:, 7, :] = 2 % n                       # a ball made of one point
    pts = rs.randn(b, n, c).astype(np.float32)
    go = rs.randn(b, m, ns, c).astype(np.float32)
    pt = T(pts, cuda).requires_grad_(True)
    pn2._lib.lib.trace = calls = []
    try:
        out = pn2.group_point(pt, T(idx, cuda))
        out.backward(T(go, cuda))
    finally:
        pn2._lib.lib.trace = None
    assert "pn2_group_point_grad_ws" in [t[0] for t in calls]
    ref = oracle.group_point_grad(pts, idx, go)
    got = pt.grad.cpu().numpy()
    assert np.allclose(got, ref, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(ref).max() * 1e-2))
    if n > 3:
        assert not got[:, 3].any()


@pytest.mark.parametrize("c,m", [(32, 300), (128, 1024), (36, 77), (512, 64), (1024, 5)])
def test_three_interpolate_grad_gather_path(pn2, oracle, cuda, c, m):
    """Levels large enough for the list-and-gather gradient (pn2_three_interpolate_grad_ws): sources nobody references
    (zero rows), one hot source, repeated indices inside a row.  Same tolerance as the atomic path."""
    import torch
    b, n = 4, 4096
    rs = np.random.RandomState(c + m)
    idx = rs.randint(0, m, (b, n, 3)).astype(np.int32)
    idx[idx == 3 % m] = 0                      # source 3 never referenced (when m > 3)
    idx[:, ::7, :] = 1 % m                     # hot source, three times per row
    w = rs.rand(b, n, 3).astype(np.float32)
    w /= w.sum(2, keepdims=True)
    pts = rs.randn(b, m, c).astype(np.float32)
    go = rs.randn(b, n, c).astype(np.float32)
    pt = T(pts, cuda).requires_grad_(True)
    pn2._lib.lib.trace = calls = []
    try:
        out = pn2.three_interpolate(pt, T(idx, cuda), T(w, cuda))
        out.backward(T(go, cuda))
    finally:
        pn2._lib.lib.trace = None
    assert "pn2_three_interpolate_grad_ws" in [t[0] for t in calls]
    ref = oracle.three_interpolate_grad(pts, idx, w, go)
    got = pt.grad.cpu().numpy()
    # both sides add ~3n/m fp32 terms per source in different orders (the oracle in index order, the kernel in list
    # order): the tolerance scales with the random-walk rounding of that many terms (m = 5: 2458 terms per source)
    atol = 1e-4 * max(1.0, np.abs(ref).max() * 1e-2) + 2e-6 * np.sqrt(3.0 * n / m) * np.abs(go).max()
    assert np.allclose(got, ref, rtol=1e-4, atol=atol), np.abs(got - ref).max()
    if m > 3:
        assert not got[:, 3].any()


def test_three_interpolate_reference_shapes(pn2, oracle, cuda):
    """tf_ops/test_tf_ops.py:59-78: seed 100; three_nn((32,512,3),(32,128,3)), weights 1/3, pts (32,128,64)."""
    np.random.seed(100)
    pts = np.random.random((32, 128, 64)).astype("float32")
    tmp1 = np.random.random((32, 512, 3)).astype("float32")
    tmp2 = np.random.random((32, 128, 3)).astype("float32")
    d, i = pn2.three_nn(T(tmp1, cuda), T(tmp2, cuda))
    rd, ri = oracle.three_nn(tmp1, tmp2)
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(d.cpu().numpy(), rd)
    w = np.ones_like(rd) / np.float32(3.0)
    out = pn2.three_interpolate(T(pts, cuda), i, T(w, cuda))
    assert np.array_equal(out.cpu().numpy(), oracle.three_interpolate(pts, ri, w))


@pytest.mark.parametrize("n,m", [(64, 16), (200, 50), (1024, 256), (8192, 1024), (20000, 33)])
def test_fps_gather_fused_equals_separate(pn2, cuda, n, m):
    """pn2_fps_gather == farthest_point_sample followed by gather_point, bit for bit."""
    import torch
    x = T(s_scene(n, 3, n), cuda)
    idx, nx = pn2.tf_ops.tf_sampling.farthest_point_sample_and_gather(m, x)
    ref_idx = pn2.farthest_point_sample(m, x)
    assert torch.equal(idx, ref_idx)
    assert torch.equal(nx, pn2.gather_point(x, ref_idx))


# ------------------------------------------------------------------ knn / selection sort
def test_select_top_k_whole_rows_bit_exact(pn2, oracle, cuda):
    """SelectionSort returns the whole (b,m,n) rows: sorted head AND the swap-permuted tail."""
    rs = np.random.RandomState(0)
    dist = rs.random_sample((3, 20, 300)).astype(np.float32)
    dist[:, :, ::7] = dist[:, :, 3:4]  # ties: the strict '<' + swaps decide the order
    for k in (1, 5, 64, 300, 400):
        oi, od = pn2.select_top_k(k, T(dist, cuda))
        ri, rd = oracle.select_top_k(k, dist)
        assert np.array_equal(oi.cpu().numpy(), ri) and np.array_equal(od.cpu().numpy(), rd), k


def test_knn_point_reference_test_shapes(pn2, oracle, cuda):
    """tf_ops/test_tf_ops.py:9-36 (knn=True branch): seed 100, xyz1 (32,512,3), xyz2 (32,128,3), k=64,
    then group_point on pts (32,512,64)."""
    np.random.seed(100)
    pts = np.random.random((32, 512, 64)).astype("float32")
    tmp1 = np.random.random((32, 512, 3)).astype("float32")
    tmp2 = np.random.random((32, 128, 3)).astype("float32")
    val, idx = pn2.knn_point(64, T(tmp1, cuda), T(tmp2, cuda))
    rv, ri = oracle.knn_point(64, tmp1, tmp2)
    assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(val.cpu().numpy(), rv)
    assert (np.diff(rv, axis=2) >= 0).all()
    g = pn2.group_point(T(pts, cuda), idx)
    assert np.array_equal(g.cpu().numpy(), oracle.group_point(pts, ri))


def test_sample_and_group_knn(pn2, oracle, cuda):
    rs = np.random.RandomState(5)
    xyz = rs.random_sample((2, 400, 3)).astype(np.float32)
    pts = rs.randn(2, 400, 4).astype(np.float32)
    nx, npts, idx, gx = pn2.sample_and_group(32, None, 8, T(xyz, cuda), T(pts, cuda), knn=True)
    f = oracle.farthest_point_sample(32, xyz)
    rnx = oracle.gather_point(xyz, f)
    _, ri = oracle.knn_point(8, xyz, rnx)
    assert np.array_equal(idx.cpu().numpy(), ri)
    assert np.array_equal(npts.cpu().numpy(),
                          np.concatenate([oracle.group_point(xyz, ri) - rnx[:, :, None], oracle.group_point(pts, ri)], -1))


# ------------------------------------------------------------------ InterpolateLabelWithColor --
def _label_case(case):
    rs = np.random.RandomState(abs(hash(case)) % 1000)
    if case == "uniform":
        sp, dp = rs.rand(5000, 3), rs.rand(20000, 3)
    elif case == "scene_surface":  # 2.5-D: almost all cells empty, a few crowded
        sp = np.concatenate([rs.uniform(-20, 20, (8000, 2)), np.abs(rs.normal(0, 0.05, (8000, 1)))], 1)
        dp = np.concatenate([rs.uniform(-21, 21, (30000, 2)), np.abs(rs.normal(0, 0.3, (30000, 1)))], 1)
    elif case == "grid_ties":  # lattice coordinates: equidistant neighbours everywhere -> lowest index wins
        sp = rs.randint(0, 12, (3000, 3)) / 4.0
        dp = rs.randint(0, 24, (10000, 3)) / 8.0
    elif case == "far_outside":  # dense points far away from the sparse cloud's bounding box
        sp = rs.rand(2000, 3) * 2
        dp = rs.rand(4000, 3) * 2
        dp[::3] += 500.0
        dp[1::7] -= np.array([300.0, 0.0, 40.0])
    elif case == "duplicates":
        base = rs.rand(40, 3)
        sp = base[rs.randint(0, 40, 4000)]
        dp = base[rs.randint(0, 40, 6000)] + (rs.rand(6000, 3) < 0.3) * 0.01
    elif case == "one_sparse":
        sp, dp = rs.rand(1, 3), rs.rand(100, 3)
    elif case == "line":  # zero extent on two axes
        sp = np.concatenate([rs.rand(1000, 1), np.zeros((1000, 2))], 1)
        dp = rs.rand(3000, 3) - 0.25
    else:
        raise KeyError(case)
    sl = rs.randint(0, 9, sp.shape[0]).astype(np.int32)
    return sp.astype(np.float32), sl, dp.astype(np.float32)


@pytest.mark.parametrize("knn", [1, 3, 4, 8])
@pytest.mark.parametrize("case", ["uniform", "scene_surface", "grid_ties", "far_outside", "duplicates", "one_sparse", "line"])
def test_interpolate_label_with_color_bit_exact(pn2, oracle, cuda, case, knn):
    """Dense-label kNN vote on the device grid == brute-force float64 oracle (labels and colours identical)."""
    sp, sl, dp = _label_case(case)
    lab, col = pn2.interpolate_label_with_color(T(sp, cuda), T(sl, cuda), T(dp, cuda), knn)
    rl, rc = oracle.interpolate_label_with_color(sp, sl, dp, knn)
    assert lab.dtype == __import__("torch").int32 and col.dtype == __import__("torch").uint8
    assert np.array_equal(lab.cpu().numpy(), rl)
    assert np.array_equal(col.cpu().numpy(), rc)


def test_interpolate_label_edge_cases(pn2, oracle, cuda):
    import torch
    dp = np.random.RandomState(0).rand(50, 3).astype(np.float32)
    # no sparse points: label -1, colour 0 (documented divergence; the reference indexes out of bounds)
    lab, col = pn2.interpolate_label_with_color(torch.zeros((0, 3), device=cuda), torch.zeros((0,), dtype=torch.int32, device=cuda),
                                                T(dp, cuda), 3)
    assert (lab.cpu().numpy() == -1).all() and not col.cpu().numpy().any()
    # labels outside the 9-entry colour table keep their label, colour 0
    sp = np.random.RandomState(1).rand(20, 3).astype(np.float32)
    sl = np.full((20,), 12, np.int32)
    lab, col = pn2.interpolate_label_with_color(T(sp, cuda), T(sl, cuda), T(dp, cuda), 3)
    rl, rc = oracle.interpolate_label_with_color(sp, sl, dp, 3)
    assert np.array_equal(lab.cpu().numpy(), rl) and np.array_equal(col.cpu().numpy(), rc) and (rl == 12).all()
    # argument errors mirror the op glue (tf_interpolate.cpp:127-160)
    with pytest.raises(ValueError, match="sparse_points must be"):
        pn2.interpolate_label_with_color(T(sp[:, :2], cuda), T(sl, cuda), T(dp, cuda), 3)
    with pytest.raises(ValueError, match="sparse_labels must be"):
        pn2.interpolate_label_with_color(T(sp, cuda), T(sl[:5], cuda), T(dp, cuda), 3)
    with pytest.raises(ValueError, match="knn must be an int scalar"):
        pn2.interpolate_label_with_color(T(sp, cuda), T(sl, cuda), T(dp, cuda), 3.0)


def test_interpolate_label_large(pn2, oracle, cuda):
    """200 k sparse / 2 M dense points (the reference runs this stage on 10^6-10^8 points): a random sample of the
    dense points is checked against the brute-force oracle, and the whole result is invariant under a permutation of
    the dense points (every query is independent)."""
    import torch
    rs = np.random.RandomState(7)
    ns, nd = 200000, 2000000
    sp = np.concatenate([rs.uniform(-50, 50, (ns, 2)), np.abs(rs.normal(0, 2.0, (ns, 1)))], 1).astype(np.float32)
    sl = rs.randint(0, 9, ns).astype(np.int32)
    dp = np.concatenate([rs.uniform(-50, 50, (nd, 2)), np.abs(rs.normal(0, 2.0, (nd, 1)))], 1).astype(np.float32)
    lab, col = pn2.interpolate_label_with_color(T(sp, cuda), T(sl, cuda), T(dp, cuda), 3)
    lab, col = lab.cpu().numpy(), col.cpu().numpy()
    pick = rs.choice(nd, 3000, replace=False)
    rl, rc = oracle.interpolate_label_with_color(sp, sl, dp[pick], 3)
    assert np.array_equal(lab[pick], rl) and np.array_equal(col[pick], rc)
    perm = rs.permutation(nd)
    lab2, _ = pn2.interpolate_label_with_color(T(sp, cuda), T(sl, cuda), T(dp[perm], cuda), 3)
    assert np.array_equal(lab2.cpu().numpy(), lab[perm])


# ------------------------------------------------------------------ multi-radius ball query (configs[2]) -----
@pytest.mark.parametrize("case", ["scene", "grid", "tiny", "dense"])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_query_ball_point_multi_equals_separate_calls(pn2, oracle, cuda, case, mode):
    """One scan for several (radius, nsample) pairs == separate query_ball_point calls == the oracle, bit for bit."""
    set_mode(mode)
    try:
        if case == "scene":
            xyz, m, radii, ks = s_scene(11, 3, 4096), 512, [0.25, 0.5, 1.0], [16, 32, 64]
        elif case == "grid":  # exact arithmetic, many points exactly on a radius
            xyz, m, radii, ks = s_grid(4, 2, 2000, 16), 200, [0.125, 0.25], [8, 32]
        elif case == "tiny":  # fewer than 8 points per segment, m not a multiple of 64
            xyz, m, radii, ks = s_randn(5, 2, 37), 13, [0.5, 1.0, 3.0], [4, 16, 48]
        else:  # every point inside every radius: early-full lists, first-nsample-in-index-order semantics
            xyz, m, radii, ks = (s_scene(12, 2, 3000) * np.float32(0.01)), 100, [1.0, 2.0, 5.0], [16, 32, 64]
        new_xyz = xyz[:, :m].copy()
        multi = pn2.tf_ops.tf_grouping.query_ball_point_multi(radii, ks, T(xyz, cuda), T(new_xyz, cuda))
        for (idx, cnt), r, k in zip(multi, radii, ks):
            si, sc = pn2.query_ball_point(r, k, T(xyz, cuda), T(new_xyz, cuda))
            oi, oc = oracle.query_ball_point(r, k, xyz, new_xyz, mode)
            assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(cnt.cpu().numpy(), oc)
            assert np.array_equal(si.cpu().numpy(), oi) and np.array_equal(sc.cpu().numpy(), oc)
    finally:
        set_mode(None)


def test_query_ball_point_multi_falls_back_when_lists_do_not_fit(pn2, oracle, cuda):
    xyz = s_scene(13, 1, 2048)
    new_xyz = xyz[:, :64].copy()
    res = pn2.tf_ops.tf_grouping.query_ball_point_multi([0.3, 0.6, 1.2], [64, 128, 128], T(xyz, cuda), T(new_xyz, cuda))
    for (idx, cnt), r, k in zip(res, [0.3, 0.6, 1.2], [64, 128, 128]):
        oi, oc = oracle.query_ball_point(r, k, xyz, new_xyz)
        assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(cnt.cpu().numpy(), oc)


# ------------------------------------------------------------------ ball query: LDS grid kernel ---------------
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("case", ["scene", "grid_ties", "dense", "tiny_radius", "huge_radius", "outside", "ragged", "flat", "k64"])
def test_query_ball_point_grid_kernel_bit_exact(pn2, oracle, cuda, case, mode):
    """The per-block LDS-grid ball query (forced with the tuning hook so that small shapes reach it too) against the
    oracle and against the scan kernel: sparse and dense neighbourhoods (rank ordering vs ordered fallback),
    lattice points exactly on the radius, radii far below / above the cloud extent, queries outside the cloud's
    bounding box, sizes that are not multiples of the tile sizes, degenerate (flat) clouds."""
    raw = pn2._lib._raw
    rs = np.random.RandomState(len(case) + mode)
    K, r = 32, 0.5
    if case == "scene":
        xyz = s_scene(21, 2, 8192); q = xyz[:, :1024].copy()
    elif case == "grid_ties":
        xyz = s_grid(22, 2, 4096, 16); q = xyz[:, :300].copy(); r = 0.25
    elif case == "dense":  # every ball holds far more than kBqgCap points -> ordered fallback
        xyz = s_scene(23, 2, 5000) * np.float32(0.05); q = xyz[:, :130].copy()
    elif case == "tiny_radius":
        xyz = s_scene(24, 1, 3000); q = xyz[:, :257].copy(); r = 1e-3
    elif case == "huge_radius":
        xyz = s_scene(25, 1, 2000); q = xyz[:, :64].copy(); r = 1e4
    elif case == "outside":  # queries up to 2 radii outside the cloud's bounding box, and some far away
        xyz = s_scene(26, 2, 4000); q = (xyz[:, :500] + rs.uniform(-1.0, 1.0, (2, 500, 3))).astype(np.float32)
        q[:, ::17] += np.float32(100.0)
    elif case == "ragged":
        xyz = s_randn(27, 3, 1237); q = s_randn(28, 3, 71); r = 0.4
    elif case == "flat":
        xyz = s_scene(29, 2, 2048); xyz[..., 2] = 0; q = xyz[:, :128].copy()
    else:  # k64
        xyz = s_scene(30, 2, 8192); q = xyz[:, :512].copy(); K, r = 64, 1.0
    set_mode(mode)
    try:
        oi, oc = oracle.query_ball_point(r, K, xyz, q, mode)
        gi, gc = pn2.query_ball_point(r, K, T(xyz, cuda), T(q, cuda), kernel=3)  # LDS grid
        si, sc = pn2.query_ball_point(r, K, T(xyz, cuda), T(q, cuda), kernel=2)  # lane-per-query scan
    finally:
        set_mode(None)
    assert np.array_equal(gc.cpu().numpy(), oc) and np.array_equal(gi.cpu().numpy(), oi)
    assert np.array_equal(sc.cpu().numpy(), oc) and np.array_equal(si.cpu().numpy(), oi)


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("case", ["sa1_scene", "n4096", "ragged_n", "dense_k64", "lattice_on_radius", "flat"])
def test_query_ball_point_binned_equals_unbinned(pn2, oracle, cuda, case, mode):
    """pn2_ball_query_bin + pn2_query_ball_point_binned (the cloud binned ONCE per cloud, query workgroups copy the bins)
    == pn2_query_ball_point == the oracle, bit for bit: SA1 shape, a cloud that is not a multiple of 64, dense balls that
    overflow the hit list (bitmap path), lattice points exactly on the radius, a degenerate axis; bins reused for two
    query sets; shapes outside the grid kernel fall back."""
    g = pn2.tf_ops.tf_grouping
    if case == "sa1_scene":
        xyz, m, r, K = s_scene(61, 16, 8192), 1024, 0.5, 32
    elif case == "n4096":
        xyz, m, r, K = s_scene(62, 3, 4096), 512, 0.7, 32
    elif case == "ragged_n":
        xyz, m, r, K = s_randn(63, 2, 5003), 300, 0.3, 16
    elif case == "dense_k64":
        xyz, m, r, K = s_scene(64, 2, 8192), 512, 1.0, 64
    elif case == "lattice_on_radius":
        xyz, m, r, K = s_grid(65, 2, 6000, 32), 256, 4 / 32.0, 32
    else:
        xyz = s_scene(66, 2, 4100); xyz[..., 2] = 0; m, r, K = 257, 0.6, 32
    q = xyz[:, :m].copy()
    set_mode(mode)
    oi, oc = oracle.query_ball_point(r, K, xyz, q, mode)
    xt = T(xyz, cuda)
    bins = g.ball_query_bin(r, xt)
    assert bins is not None
    bi, bc = g.query_ball_point_binned(r, K, xt, T(q, cuda), bins)
    assert np.array_equal(bi.cpu().numpy(), oi) and np.array_equal(bc.cpu().numpy(), oc)
    # bins that do not describe the query are refused (the C entry point takes an opaque pointer)
    with pytest.raises(ValueError, match="do not describe"):
        g.query_ball_point_binned(r * 0.5, K, xt, T(q, cuda), bins)
    with pytest.raises(ValueError, match="do not describe"):
        g.query_ball_point_binned(r, K, T(xyz[:1], cuda), T(q[:1], cuda), bins)
    # ... and so are bins of ANOTHER cloud of the same shape, or of a cloud modified in place since
    with pytest.raises(ValueError, match="another cloud"):
        g.query_ball_point_binned(r, K, T(xyz, cuda), T(q, cuda), bins)
    q2 = xyz[:, -m:].copy()  # the bins depend on (radius, xyz1) only
    b2, c2 = g.query_ball_point_binned(r, K, xt, T(q2, cuda), bins)
    o2, oc2 = oracle.query_ball_point(r, K, xyz, q2, mode)
    assert np.array_equal(b2.cpu().numpy(), o2) and np.array_equal(c2.cpu().numpy(), oc2)
    # bins of a column block of a wider batch read in place (pn2_ball_query_bin_ld) are the bins of the dense copy, bit for bit,
    # and the binned query takes the strided view as it is (model.sa1_samples hands point_cloud[:, :, 0:3] over)
    import torch
    wide = torch.cat([xt, torch.rand_like(xt)], dim=2)      # (b, n, 6): xyz | colours
    view = wide[:, :, 0:3]
    bins_v = g.ball_query_bin(r, view)
    bv, cv = g.query_ball_point_binned(r, K, view, T(q, cuda), bins_v)
    assert torch.equal(bv, bi) and torch.equal(cv, bc)
    assert g.ball_query_bin(r, T(xyz[:, :1000].copy(), cuda)) is None  # below the grid kernel's range: caller scans
    xt.add_(0.0)  # an in-place update (same values): the tag no longer vouches for the cloud
    with pytest.raises(ValueError, match="another cloud"):
        g.query_ball_point_binned(r, K, xt, T(q, cuda), bins)


def test_query_ball_point_kernels_fuzz(pn2, oracle, cuda):
    """Differential fuzz: 40 random (cloud, queries, radius, nsample) draws, every ball-query kernel (LDS grid, lane
    scan, wave-per-queries scan) against the oracle.  Mixes float clouds, lattice clouds (hits exactly on the
    radius and on cell boundaries), anisotropic extents and radii from 1/200 to 2x of the extent."""
    raw = pn2._lib._raw
    rs = np.random.RandomState(1234)
    try:
        for it in range(40):
            b = int(rs.randint(1, 4))
            n = int(rs.choice([64, 333, 1024, 2500, 4096, 8192]))
            m = int(rs.choice([1, 17, 64, 300, 1024]))
            K = int(rs.choice([1, 8, 16, 32, 64]))
            ext = rs.choice([0.01, 1.0, 10.0, 1000.0], 3).astype(np.float32)
            if it % 3 == 0:  # lattice: many exact ties
                g = int(rs.choice([8, 16, 64]))
                xyz = (rs.randint(0, g, (b, n, 3)) / np.float32(g)).astype(np.float32) * ext
            else:
                xyz = (rs.random_sample((b, n, 3)).astype(np.float32) - np.float32(0.5 * (it % 2))) * ext
            if it % 4 == 1:
                q = (rs.random_sample((b, m, 3)).astype(np.float32) * 1.2 - 0.1).astype(np.float32) * ext
            else:
                q = xyz[:, rs.randint(0, n, m)].copy()
            r = float(ext.max() * rs.choice([0.005, 0.02, 0.08, 0.3, 2.0]))
            if it % 3 == 0 and it % 2 == 0:
                r = float(ext.max() * rs.choice([1, 2, 4]) / g)  # a lattice distance: sqrt(d2) == radius cases
            oi, oc = oracle.query_ball_point(r, K, xyz, q)
            for variant in (3, 2, 1):
                gi, gc = pn2.query_ball_point(r, K, T(xyz, cuda), T(q, cuda), kernel=variant)
                assert np.array_equal(gc.cpu().numpy(), oc), (it, variant, n, m, K, r)
                assert np.array_equal(gi.cpu().numpy(), oi), (it, variant, n, m, K, r)
    finally:
        pass


def test_three_nn_fuzz(pn2, oracle, cuda):
    """Differential fuzz of three_nn's fp32 ranking filter: random sizes, extents from 1e-3 to 1e4, offsets up to 1e5
    (where the fp32 grid becomes coarser than the point spacing), lattices and duplicated points."""
    rs = np.random.RandomState(4321)
    for it in range(30):
        b = int(rs.randint(1, 3))
        n = int(rs.choice([1, 63, 500, 2000]))
        m = int(rs.choice([3, 4, 64, 65, 700, 1024, 1500]))
        ext = (rs.choice([1e-3, 1.0, 30.0, 1e4], 3)).astype(np.float32)
        off = (rs.choice([0.0, 0.0, 5.0, 1e3, 1e5], 3)).astype(np.float32)
        if it % 3 == 0:
            g = int(rs.choice([4, 16]))
            r = (rs.randint(0, g, (b, m, 3)) / np.float32(g)).astype(np.float32) * ext + off
            a = (rs.randint(0, 2 * g, (b, n, 3)) / np.float32(2 * g)).astype(np.float32) * ext + off
        else:
            r = rs.random_sample((b, m, 3)).astype(np.float32) * ext + off
            a = rs.random_sample((b, n, 3)).astype(np.float32) * ext + off
        if it % 5 == 2:
            r[:, ::2] = r[:, 1::2][:, : r[:, ::2].shape[1]] if m % 2 == 0 else r[:, ::2]  # duplicated known points
        a, r = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(r, np.float32)
        d, i = pn2.three_nn(T(a, cuda), T(r, cuda))
        rd, ri = oracle.three_nn(a, r)
        assert np.array_equal(i.cpu().numpy(), ri), (it, n, m)
        assert np.array_equal(d.cpu().numpy(), rd), (it, n, m)


# ------------------------------------------------------------------ prob_sample ----------------
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 100, 4095, 8192, 8193, 16384, 20001, 70000])
def test_prob_sample_bit_exact(pn2, oracle, cuda, n):
    """ProbSample: the running sums (read back through the C ABI) and the drawn indices equal the oracle's restatement
    of the reference's summation order bit for bit, across chunk boundaries (8192) and ragged tails."""
    import ctypes
    import torch
    rs = np.random.RandomState(n)
    b, m = 3, 777
    inp = rs.rand(b, n).astype(np.float32)
    inp[0, : n // 2] = 0.0  # runs of zero weight: equal running sums, the search must still agree
    inpr = rs.rand(b, m).astype(np.float32)
    inpr[:, 0] = 0.0
    inpr[:, 1] = np.float32(1.0) - np.float32(2 ** -24)
    out = pn2.prob_sample(T(inp, cuda), T(inpr, cuda))
    ro, rcs = oracle.prob_sample(inp, inpr)
    assert np.array_equal(out.cpu().numpy(), ro)
    temp = torch.empty((b, n), dtype=torch.float32, device=cuda)
    o2 = torch.empty((b, m), dtype=torch.int32, device=cuda)
    L = pn2._lib.lib
    assert L.pn2_prob_sample(b, n, m, ctypes.c_void_p(T(inp, cuda).data_ptr()), ctypes.c_void_p(T(inpr, cuda).data_ptr()),
                             ctypes.c_void_p(temp.data_ptr()), ctypes.c_void_p(o2.data_ptr()), None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(temp.cpu().numpy(), rcs)
    # and it is a sampler: the empirical distribution follows the weights
    if n == 100:
        w = rs.rand(1, n).astype(np.float32)
        draws = pn2.prob_sample(T(w, cuda), T(rs.rand(1, 200000).astype(np.float32), cuda)).cpu().numpy()[0]
        freq = np.bincount(draws, minlength=n) / 200000.0
        assert np.abs(freq - w[0] / w[0].sum()).max() < 4e-3


def test_kernels_against_frozen_extra_fixtures(pn2, cuda):
    """The HIP kernels against tests/golden/oracle_extra.npz (frozen oracle outputs: the oracle library itself is not
    called here): selection sort / kNN, label interpolation, prob_sample, bf16 rounding of the device."""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    x = mg.extra_inputs()
    g = np.load(os.path.join(GOLD, "oracle_extra.npz"))
    oi, ov = pn2.select_top_k(7, T(x["dist"], cuda))
    assert np.array_equal(oi.cpu().numpy(), g["topk_idx"]) and np.array_equal(ov.cpu().numpy(), g["topk_val"])
    kv, ki = pn2.knn_point(5, T(x["kx1"], cuda), T(x["kx2"], cuda))
    assert np.array_equal(ki.cpu().numpy(), g["knn_idx"]) and np.array_equal(kv.cpu().numpy(), g["knn_val"])
    for k in (1, 3, 8):
        lab, col = pn2.interpolate_label_with_color(T(x["sp"], cuda), T(x["sl"], cuda), T(x["dp"], cuda), k)
        assert np.array_equal(lab.cpu().numpy(), g["label_k%d" % k]) and np.array_equal(col.cpu().numpy(), g["color_k%d" % k])
    assert np.array_equal(pn2.prob_sample(T(x["pw"], cuda), T(x["pr"], cuda)).cpu().numpy(), g["prob_idx"])
    assert np.array_equal(T(x["bf"], cuda).to(torch.bfloat16).float().cpu().numpy(), g["bf16"])
    loss = pn2.model.get_loss(T(x["logits"], cuda), T(x["labels"].astype(np.int64), cuda), T(x["smpw"], cuda))
    assert abs(float(loss) - float(g["ce"])) < 1e-5


# ------------------------------------------------------------------ FPS for large clouds (Morton buckets) ----
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("case", ["scene_20000", "scene_65536", "grid_ties", "clustered", "duplicates", "ragged_batch", "line"])
def test_fps_large_bucket_kernel_bit_exact(pn2, oracle, cuda, case, mode):
    """pn2_fps_large (Morton buckets + bounding-box skipping) picks exactly what the reference algorithm picks
    (oracle) and what the streaming kernel picks -- including lattice clouds where almost every round is decided
    by the (k mod 512, k) tie-break -- and returns the picked coordinates."""
    sm = pn2.tf_ops.tf_sampling
    rs = np.random.RandomState(len(case))
    if case == "scene_20000":
        xyz, m = s_scene(41, 1, 20000), 700
    elif case == "scene_65536":
        xyz, m = s_scene(42, 1, 65536), 1024
    elif case == "grid_ties":
        xyz, m = s_grid(43, 1, 40000, 32), 600
    elif case == "clustered":  # tight clusters far apart: most buckets are skipped from the start
        c = rs.uniform(-50, 50, (1, 40, 3))
        xyz = (c[:, rs.randint(0, 40, 30000)] + rs.normal(0, 0.05, (1, 30000, 3))).astype(np.float32); m = 500
    elif case == "duplicates":
        base = rs.rand(1, 300, 3).astype(np.float32)
        xyz = base[:, rs.randint(0, 300, 17000)]; m = 350  # more picks than distinct points: td reaches 0 everywhere
    elif case == "ragged_batch":
        xyz, m = s_scene(44, 3, 16385 + 37), 300
    else:  # line: degenerate bounding box on two axes
        xyz = np.zeros((1, 18000, 3), np.float32); xyz[..., 0] = rs.rand(1, 18000); m = 400
    xyz = np.ascontiguousarray(xyz, np.float32)
    set_mode(mode)
    try:
        ref = oracle.farthest_point_sample(m, xyz, mode)
        idx, new_xyz = sm.farthest_point_sample_and_gather(m, T(xyz, cuda))
        only_idx = pn2.farthest_point_sample(m, T(xyz, cuda))
        sm.USE_BUCKET_FPS = False
        try:
            streamed = pn2.farthest_point_sample(m, T(xyz, cuda))
        finally:
            sm.USE_BUCKET_FPS = True
    finally:
        set_mode(None)
    assert np.array_equal(idx.cpu().numpy(), ref)
    assert np.array_equal(only_idx.cpu().numpy(), ref)
    assert np.array_equal(streamed.cpu().numpy(), ref)
    assert np.array_equal(new_xyz.cpu().numpy(), oracle.gather_point(xyz, ref))
