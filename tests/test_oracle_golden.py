"""CPU tests: the oracle against the reference's known-answer vector and the frozen fixtures."""
import os

import numpy as np
import pytest

from conftest import s_grid, s_scene

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_three_nn_reference_golden_vector(oracle):
    """tf_ops/test_interpolate.py:30-35 of the reference -- its only forward KAT for this path."""
    g = np.load(os.path.join(GOLD, "reference_three_nn.npz"))
    np.random.seed(int(g["seed"]))
    target = np.random.random(tuple(g["target_shape"])).astype("float32")
    reference = np.random.random(tuple(g["reference_shape"])).astype("float32")
    dist, idx = oracle.three_nn(target[:3], reference[:3])  # the KAT only inspects [:3, :3, :1]
    got_d = dist[:3, :3, :1].flatten()
    got_i = idx[:3, :3, :1].flatten()
    assert (got_i == g["idx"]).all()
    # the reference prints with numpy's default 8 significant digits
    assert np.abs(got_d.astype(np.float64) - g["dist"]).max() < 5.1e-9
    assert np.array2string(got_d) == str(g["printed"])
    assert dist.dtype == np.float32 and idx.dtype == np.int32


def test_three_nn_matches_kdtree_float64(oracle):
    """Independent check of the fp64 semantics with scipy's exact KD-tree."""
    from scipy.spatial import cKDTree
    rs = np.random.RandomState(1)
    a = rs.random_sample((2, 500, 3)).astype(np.float32)
    r = rs.random_sample((2, 64, 3)).astype(np.float32)
    dist, idx = oracle.three_nn(a, r)
    for b in range(2):
        d, i = cKDTree(r[b].astype(np.float64)).query(a[b].astype(np.float64), k=3)
        assert (i == idx[b]).all()
        assert np.array_equal((d ** 2).astype(np.float32), dist[b]) or np.allclose((d ** 2), dist[b], rtol=3e-7, atol=0)


def test_frozen_fixtures(oracle):
    g = np.load(os.path.join(GOLD, "oracle_small.npz"))
    rs = np.random.RandomState(7)
    xyz = rs.random_sample((2, 1024, 3)).astype(np.float32)
    feat = rs.random_sample((2, 1024, 3)).astype(np.float32)
    for mode in (0, 1, 2):
        f = oracle.farthest_point_sample(256, xyz, mode)
        assert (f == g["cfg0_fps_m%d" % mode]).all()
        idx, cnt = oracle.query_ball_point(0.2, 16, xyz, oracle.gather_point(xyz, f), mode)
        assert (idx == g["cfg0_bq_idx_m%d" % mode]).all() and (cnt == g["cfg0_bq_cnt_m%d" % mode]).all()
    gr = s_grid(3, 2, 1500, 64)
    for mode in (0, 1, 2):  # exact arithmetic: all modes identical
        fg = oracle.farthest_point_sample(200, gr, mode)
        assert (fg == g["grid_fps"]).all()
        gi, gc = oracle.query_ball_point(0.25, 32, gr, oracle.gather_point(gr, fg), mode)
        assert (gi == g["grid_bq_idx"]).all() and (gc == g["grid_bq_cnt"]).all()
    sc = s_scene(5, 1, 2048)
    fs = oracle.farthest_point_sample(256, sc)
    assert (fs == g["scene_fps"]).all()
    f = g["cfg0_fps_m1"]
    d, i3 = oracle.three_nn(xyz, oracle.gather_point(xyz, f))
    assert np.array_equal(d, g["cfg0_nn_dist"]) and (i3 == g["cfg0_nn_idx"]).all()
    out = oracle.three_interpolate(oracle.gather_point(feat, f), i3, oracle.fp_weights(d))
    assert np.array_equal(out, g["cfg0_interp"])


def _fps_by_key(x, m):
    """FPS with the tie-break written as an explicit sort key (max dist, k mod 512, k): the
    closed form of what the 512-thread block + tree of tf_sampling.cu:153-170 computes."""
    n = len(x)
    md = np.full(n, 1e38, np.float32)
    k = np.arange(n)
    key = (k % 512) * (1 << 22) + k // 512
    out, old = [0], 0
    for _ in range(1, m):
        dd = (x - x[old]) ** 2
        d = ((dd[:, 0] + dd[:, 1]) + dd[:, 2]).astype(np.float32)
        md = np.minimum(md, d)
        c = np.where(md == md.max())[0]
        old = int(c[np.argmin(key[c])])
        out.append(old)
    return np.array(out)


def test_fps_tie_break_closed_form(oracle):
    """The block emulation equals the closed-form key on tie-heavy grid inputs (incl. duplicates)."""
    x = s_grid(0, 3, 1300, 16)  # 4096 distinct positions for 1300 points -> many duplicates / ties
    f = oracle.farthest_point_sample(150, x, 0)
    for b in range(3):
        assert (_fps_by_key(x[b], 150) == f[b]).all()
    assert f[:, 0].tolist() == [0, 0, 0]


def test_fps_all_points_identical(oracle):
    x = np.ones((1, 700, 3), np.float32)
    f = oracle.farthest_point_sample(5, x)
    assert f.tolist() == [[0, 0, 0, 0, 0]]  # all distances 0: thread 0 / k=0 wins every round


def test_ball_query_semantics(oracle):
    rs = np.random.RandomState(2)
    xyz = rs.random_sample((1, 300, 3)).astype(np.float32)
    q = xyz[:, :10].copy()
    idx, cnt = oracle.query_ball_point(0.15, 8, xyz, q, 0)
    for j in range(10):
        d = np.sqrt(((xyz[0] - q[0, j]) ** 2).sum(1, dtype=np.float32)).astype(np.float32)
        hits = np.where(d < np.float32(0.15))[0]
        c = min(len(hits), 8)
        assert cnt[0, j] == c and c >= 1  # a query that is a dataset point always finds itself
        assert (idx[0, j, :c] == hits[:c]).all()
        assert (idx[0, j, c:] == hits[0]).all()  # padded with the FIRST hit
    # a far-away query: empty ball -> count 0, row zero-filled (documented divergence)
    far = np.full((1, 1, 3), 100.0, np.float32)
    idx, cnt = oracle.query_ball_point(0.15, 8, xyz, far, 0)
    assert cnt[0, 0] == 0 and (idx == 0).all()


def test_group_and_grads_small(oracle):
    rs = np.random.RandomState(3)
    pts = rs.random_sample((2, 20, 5)).astype(np.float32)
    idx = rs.randint(0, 20, (2, 4, 6)).astype(np.int32)
    out = oracle.group_point(pts, idx)
    for b in range(2):
        assert np.array_equal(out[b], pts[b][idx[b]])
    go = rs.random_sample(out.shape).astype(np.float32)
    gp = oracle.group_point_grad(pts, idx, go)
    ref = np.zeros_like(pts, dtype=np.float64)
    for b in range(2):
        np.add.at(ref[b], idx[b].reshape(-1), go[b].reshape(-1, 5))
    assert np.allclose(gp, ref, rtol=1e-6, atol=1e-6)
    w = rs.random_sample((2, 7, 3)).astype(np.float32)
    i3 = rs.randint(0, 20, (2, 7, 3)).astype(np.int32)
    o = oracle.three_interpolate(pts, i3, w)
    ref = sum(pts[np.arange(2)[:, None], i3[:, :, t]] * w[:, :, t:t + 1] for t in range(3))
    assert np.allclose(o, ref, rtol=1e-6, atol=1e-6)


def test_mlp_oracle_bn_fold(oracle):
    """conv_bn_relu == the folded (W', b') form the HIP path consumes."""
    rs = np.random.RandomState(4)
    x = rs.randn(50, 7)
    layer = dict(W=rs.randn(7, 5), b=rs.randn(5), gamma=rs.rand(5) + 0.5, beta=rs.randn(5), mean=rs.randn(5),
                 var=rs.rand(5) + 0.1)
    y = oracle.conv_bn_relu(x, layer)
    s = layer["gamma"] / np.sqrt(layer["var"] + 1e-3)
    y2 = np.maximum(x @ (layer["W"] * s) + ((layer["b"] - layer["mean"]) * s + layer["beta"]), 0)
    assert np.allclose(y, y2, rtol=1e-12, atol=1e-12)


def test_weighted_sparse_ce_known_answers(oracle):
    """model.py:152-161 restatement: uniform logits -> log(C); zero weights drop out of the denominator."""
    pred = np.zeros((2, 5, 9), np.float32)
    label = np.arange(10).reshape(2, 5) % 9
    w = np.ones((2, 5), np.float32)
    assert abs(oracle.weighted_sparse_ce(pred, label, w) - np.log(9.0)) < 1e-12
    w[0, :3] = 0.0
    w[1, :] = 2.0
    # sum(w*ce)/count(w!=0) = (2 + 5*2) * log 9 / 7
    assert abs(oracle.weighted_sparse_ce(pred, label, w) - 12.0 * np.log(9.0) / 7.0) < 1e-12
    assert oracle.weighted_sparse_ce(pred, label, np.zeros((2, 5))) == 0.0


def test_frozen_fixtures_extra(oracle):
    """Regression pins of the oracle functions added later (selection sort / kNN, label interpolation, prob_sample,
    loss, bf16 rounding): tests/golden/oracle_extra.npz, inputs from make_golden.extra_inputs()."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    g = np.load(os.path.join(GOLD, "oracle_extra.npz"))
    fresh = mg.extra_fixtures()
    assert sorted(fresh) == sorted(g.files)
    for k in g.files:
        assert np.array_equal(np.asarray(fresh[k]), g[k]), k


@pytest.mark.parametrize("relu", [True, False])
def test_batch_norm_train_oracle_matches_torch_autograd(oracle, relu):
    """oracle.batch_norm_relu_train(_grad) (tf_util.py:555-581 restated) against torch float64: F.batch_norm in
    training mode (same fused-BN conventions: biased variance to normalise, unbiased into the moving average) + relu,
    and autograd for the three gradients."""
    import torch
    import torch.nn.functional as F
    rs = np.random.RandomState(11)
    y = rs.randn(200, 12) * 2 + rs.randn(12)
    gamma, beta, bias, dz = rs.rand(12) + 0.5, rs.randn(12) * 0.3, rs.randn(12), rs.randn(200, 12)
    z, mean, var, mm, mv = oracle.batch_norm_relu_train(y, gamma, beta, relu, bias=bias, moving=(np.full(12, 0.25), np.full(12, 2.0)),
                                                        decay=0.9)
    ty = torch.tensor(y + bias, requires_grad=True)  # torch sees the biased output; BN removes the constant again
    tg, tb = torch.tensor(gamma, requires_grad=True), torch.tensor(beta, requires_grad=True)
    rm, rv = torch.full((12,), 0.25, dtype=torch.float64), torch.full((12,), 2.0, dtype=torch.float64)
    tz = F.batch_norm(ty, rm, rv, tg, tb, training=True, momentum=0.1, eps=1e-3)
    if relu:
        tz = torch.relu(tz)
    assert np.allclose(z, tz.detach().numpy(), rtol=1e-10, atol=1e-10)
    assert np.allclose(mm, rm.numpy(), rtol=1e-12) and np.allclose(mv, rv.numpy(), rtol=1e-12)
    (tz * torch.tensor(dz)).sum().backward()
    dy, dgamma, dbeta = oracle.batch_norm_relu_train_grad(y, gamma, beta, dz, relu)
    assert np.allclose(dy, ty.grad.numpy(), rtol=1e-9, atol=1e-10)
    assert np.allclose(dgamma, tg.grad.numpy(), rtol=1e-9, atol=1e-10)
    assert np.allclose(dbeta, tb.grad.numpy(), rtol=1e-9, atol=1e-10)


def test_max_pool_rows_oracle_matches_torch_amax(oracle):
    """oracle.max_pool_rows(_grad) (tf.reduce_max over K, pointnet_util.py:167-170, gradient shared among tied rows)
    against torch.amax and its autograd, with exact ties (duplicated rows, the ReLU floor)."""
    import torch
    rs = np.random.RandomState(3)
    z = np.maximum(rs.randn(6, 8, 5), 0.0)      # (groups, pool, c); ReLU floor ties
    z[1, 0, :] = 10.0
    z[1, 4:, :] = z[1, :1, :]                   # duplicated neighbours: the maximum is attained five times
    z[2] = 0.0                                   # a group that never passes the ReLU
    dzp = rs.randn(6, 5)
    zmax, ties = oracle.max_pool_rows(z.reshape(48, 5), 8)
    tz = torch.tensor(z, requires_grad=True)
    tm = tz.amax(dim=1)
    assert np.array_equal(zmax, tm.detach().numpy())
    assert np.array_equal(ties, (z == z.max(axis=1, keepdims=True)).sum(axis=1))
    assert ties[2].min() == 8 and ties[1].max() >= 5
    (tm * torch.tensor(dzp)).sum().backward()
    g = oracle.max_pool_rows_grad(z.reshape(48, 5), 8, dzp)
    assert np.allclose(g.reshape(6, 8, 5), tz.grad.numpy(), rtol=1e-12, atol=1e-12)


def test_three_interpolate_restatement_equals_the_lifted_reference_functions(oracle):
    """oracle/_ref/libpn2_ref_interp.so holds the reference's OWN threeinterpolate_cpu /
    threeinterpolate_grad_cpu (tf_interpolate.cpp:307-330,397-421, cut out by oracle/lift_interpolate.py at build time and
    compiled with the reference's host flags).  The C restatement must equal them bit for bit -- forward AND gradient (the
    gradient is a sequential += loop: same order, same bits) -- at the FP shapes of configs[1] scaled to CPU seconds and on
    the reference's own test shape (test_tf_ops.py:80-94: (1,8,16) points -> (1,128,16))."""
    from oracle import ref as R
    if not R.interp_available():
        pytest.skip("oracle/_ref/libpn2_ref_interp.so not built (needs /root/reference: `make -C oracle _ref`)")
    rs = np.random.RandomState(5)
    for b, m, c, n in [(1, 8, 16, 128), (2, 16, 512, 64), (3, 64, 256, 256), (2, 256, 256, 1024), (2, 1024, 128, 8192), (1, 5, 3, 7)]:
        pts = rs.randn(b, m, c).astype(np.float32)
        idx = rs.randint(0, m, (b, n, 3)).astype(np.int32)
        idx[:, ::7, 1] = idx[:, ::7, 0]  # duplicated neighbours (three_nn of duplicated points)
        d = rs.rand(b, n, 3).astype(np.float32) + 1e-3
        w = ((1.0 / d) / (1.0 / d).sum(2, keepdims=True)).astype(np.float32)  # pointnet_util.py:300-303
        go = rs.randn(b, n, c).astype(np.float32)
        assert np.array_equal(oracle.three_interpolate(pts, idx, w), R.three_interpolate(pts, idx, w))
        assert np.array_equal(oracle.three_interpolate_grad(pts, idx, w, go), R.three_interpolate_grad(pts, idx, w, go))
