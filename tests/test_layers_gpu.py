"""GPU parity tests of the MFMA layer kernels and the layer API against the numpy/fp64 oracle.
Float tolerance (north_star): fp32 MLP features within 1e-5 -> |hip - ref64| <= 1e-5 + 1e-5*|ref64|."""
import numpy as np
import pytest

from conftest import s_grid, s_randn, s_scene

pytestmark = pytest.mark.gpu
RTOL = ATOL = 1e-5


def T(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def close(got, ref):
    got = np.asarray(got, np.float64)
    err = np.abs(got - ref)
    tol = ATOL + RTOL * np.abs(ref)
    assert (err <= tol).all(), "max err %.3e (tol %.1e) at %s, ref scale %.3f" % (
        err.max(), tol.flat[err.argmax()], np.unravel_index(err.argmax(), err.shape), np.abs(ref).max())


def randomize_bn(store, seed):
    """non-trivial BN statistics so that folding is actually exercised"""
    import torch
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, v in store.params.items():
            if k.endswith("bn/gamma"):
                v.copy_((torch.rand(v.shape, generator=g) + 0.5).to(v.device))
            elif k.endswith("bn/beta") or k.endswith("biases"):
                v.copy_((torch.randn(v.shape, generator=g) * 0.1).to(v.device))
        for k, v in store.buffers.items():
            if k.endswith("moving_mean"):
                v.copy_((torch.randn(v.shape, generator=g) * 0.1).to(v.device))
            elif k.endswith("moving_variance"):
                v.copy_((torch.rand(v.shape, generator=g) + 0.5).to(v.device))


def layer_dicts(store, scope, names, bn=True):
    out = []
    for nm in names:
        p = ("%s/" % scope if scope else "") + nm + "/"
        W = store.params[p + "weights"].detach().cpu().numpy().astype(np.float64)
        W = W.reshape(W.shape[-2], W.shape[-1])
        d = dict(W=W, b=store.params[p + "biases"].detach().cpu().numpy())
        if bn:
            d.update(gamma=store.params[p + "bn/gamma"].detach().cpu().numpy(),
                     beta=store.params[p + "bn/beta"].detach().cpu().numpy(),
                     mean=store.buffers[p + "bn/moving_mean"].cpu().numpy(),
                     var=store.buffers[p + "bn/moving_variance"].cpu().numpy())
        out.append(d)
    return out


# ------------------------------------------------------------------ pn2_linear --------------
@pytest.mark.parametrize("rows,cin,cout", [(128, 16, 32), (100, 6, 32), (257, 67, 64), (512, 131, 128), (300, 259, 256),
                                           (1024, 768, 256), (64, 128, 512), (4096, 128, 128)])
@pytest.mark.parametrize("relu", [0, 1])
def test_linear_vs_fp64(pn2, cuda, rows, cin, cout, relu):
    rs = np.random.RandomState(rows + cin)
    x = rs.randn(rows, cin).astype(np.float32)
    w = (rs.randn(cin, cout) / np.sqrt(cin)).astype(np.float32)
    b = rs.randn(cout).astype(np.float32)
    y = pn2.util.tf_util.hip_linear(T(x, cuda), T(w, cuda), T(b, cuda), relu=relu).cpu().numpy()
    ref = x.astype(np.float64) @ w.astype(np.float64) + b
    if relu:
        ref = np.maximum(ref, 0)
    close(y, ref)


@pytest.mark.parametrize("rows,cin,cout", [(70000, 134, 128), (66000, 67, 256), (131072, 131, 128)])
def test_linear_large_rows_scalar_a_loads(pn2, cuda, rows, cin, cout):
    """Many rows with cin % 4 != 0 (the reference's [xyz | features] widths 67 / 131 / 134): the 128-row tile has no
    scalar-load variant (its accumulators spilled to scratch: 1.7 ms instead of 55 us), so these shapes must take the
    64-row tile -- numerics against float64 and a loose guard on the time."""
    import torch
    rs = np.random.RandomState(rows % 1000 + cin)
    x = rs.randn(rows, cin).astype(np.float32)
    w = (rs.randn(cin, cout) / np.sqrt(cin)).astype(np.float32)
    b = rs.randn(cout).astype(np.float32)
    tx, tw, tb = T(x, cuda), T(w, cuda), T(b, cuda)
    y = pn2.util.tf_util.hip_linear(tx, tw, tb, relu=1)
    close(y.cpu().numpy(), np.maximum(x.astype(np.float64) @ w.astype(np.float64) + b, 0))
    best = 1e9
    for _ in range(4):  # best of four: a timing guard must not trip on one slow sample (clock ramp, neighbours)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            pn2.util.tf_util.hip_linear(tx, tw, tb, relu=1)
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 5)
    assert best < 0.8, "pn2_linear %.3f ms per call: a spilling tile configuration? (1.7 ms when it spilled)" % best


def test_linear_layout_is_transpose_detecting(pn2, cuda):
    """A = identity-like, asymmetric W: catches swapped row/col or k-slice mix-ups exactly."""
    cin = cout = 64
    x = np.zeros((128, cin), np.float32)
    for r in range(128):
        x[r, (r * 7) % cin] = 1.0
    w = (np.arange(cin * cout, dtype=np.float32).reshape(cin, cout) % 251) + np.arange(cout, dtype=np.float32) * 0.5
    y = pn2.util.tf_util.hip_linear(T(x, cuda), T(w, cuda), T(np.zeros(cout, np.float32), cuda), relu=0).cpu().numpy()
    assert np.array_equal(y, x @ w)


@pytest.mark.parametrize("pool", [16, 32, 64])
@pytest.mark.parametrize("cout", [32, 64, 128, 256])
def test_linear_fused_maxpool(pn2, cuda, pool, cout):
    rs = np.random.RandomState(pool + cout)
    rows, cin = pool * 37, 40
    x = rs.randn(rows, cin).astype(np.float32)
    w = (rs.randn(cin, cout) / 6).astype(np.float32)
    b = rs.randn(cout).astype(np.float32)
    y = pn2.util.tf_util.hip_linear(T(x, cuda), T(w, cuda), T(b, cuda), relu=1, pool=pool).cpu().numpy()
    ref = np.maximum(x.astype(np.float64) @ w.astype(np.float64) + b, 0).reshape(37, pool, cout).max(1)
    assert y.shape == (37, cout)
    close(y, ref)


# ------------------------------------------------------------------ SA module ---------------
SA_CASES = [
    # (name, B, N, C, npoint, radius, mlp)                      fused kernel instantiation
    ("sa1-like", 2, 1024, 3, 128, 0.25, [32, 32, 64]),       # <3,1,1,2,scalar>
    ("sa2-like", 2, 512, 64, 64, 0.4, [64, 64, 128]),         # <3,2,2,4,vec8>
    ("ns-shape", 2, 512, 128, 64, 0.4, [128]),                # <1,4,vec8>  north-star layer
    ("no-feat", 1, 700, 0, 50, 0.3, [32, 32, 64]),            # points=None
    ("msg-mid", 1, 512, 16, 32, 0.4, [64, 96, 128]),          # <3,2,3,4,vec8>
    ("wide", 1, 256, 131, 32, 0.5, [128, 128, 256]),          # unsupported by fused -> group_concat + linear
]


@pytest.mark.parametrize("case", SA_CASES, ids=[c[0] for c in SA_CASES])
@pytest.mark.parametrize("fused", [True, False])
def test_sa_module_inference_vs_oracle(pn2, oracle, cuda, case, fused):
    name, B, N, C, npoint, radius, mlp = case
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    rs = np.random.RandomState(len(name) + N)
    xyz = rs.random_sample((B, N, 3)).astype(np.float32)
    pts = rs.randn(B, N, C).astype(np.float32) if C else None
    store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=1))
    pu.USE_FUSED_SA = fused
    try:
        args = (T(xyz, cuda), T(pts, cuda) if C else None)
        kw = dict(npoint=npoint, radius=radius, nsample=32, mlp=mlp, mlp2=None, group_all=False, is_training=False,
                  bn_decay=None, scope="sa")
        pu.pointnet_sa_module(*args, **kw)  # creates the variables
        randomize_bn(store, 3)
        new_xyz, new_points, idx = pu.pointnet_sa_module(*args, **kw)
    finally:
        pu.USE_FUSED_SA = True
    layers = layer_dicts(store, "sa", ["conv%d" % i for i in range(len(mlp))])
    r_xyz, r_pts, r_idx = oracle.sa_module(xyz, pts, npoint, radius, 32, layers)
    assert np.array_equal(idx.cpu().numpy(), r_idx)            # FPS + ball query: bit exact
    assert np.array_equal(new_xyz.cpu().numpy(), r_xyz)
    assert new_points.shape == (B, npoint, mlp[-1])
    close(new_points.cpu().numpy(), r_pts)


@pytest.mark.parametrize("K,npoint,c,mlp", [(16, 64, 3, [32, 32, 64]), (64, 96, 64, [64, 64, 128]), (128, 50, 8, [128]),
                                            (16, 33 * 2, 0, [32, 32, 64]), (64, 7, 3, [64, 128])])
def test_sa_fused_other_neighbourhood_sizes(pn2, oracle, cuda, K, npoint, c, mlp):
    """The fused gather+MLP+max kernel for K != 32: two centres per 32-row tile (K=16) or several tiles per centre
    merged with atomicMax (K=64,128), against the oracle."""
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    rs = np.random.RandomState(K + npoint)
    xyz = s_scene(K, 2, 1500)
    pts = rs.randn(2, 1500, c).astype(np.float32) if c else None
    store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=K))
    args = (T(xyz, cuda), None if pts is None else T(pts, cuda), npoint, 1.2, K, mlp, None, False, False, None)
    pu.pointnet_sa_module(*args, scope="sa")
    randomize_bn(store, K + 1)
    calls = []
    pn2._lib.lib.trace = calls
    try:
        new_xyz, new_points, idx = pu.pointnet_sa_module(*args, scope="sa")
    finally:
        pn2._lib.lib.trace = None
    assert "pn2_sa_mlp_max_fused" in [c_[0] for c_ in calls] and "pn2_linear" not in [c_[0] for c_ in calls]
    layers = layer_dicts(store, "sa", ["conv%d" % i for i in range(len(mlp))])
    rx, rp, ri = oracle.sa_module(xyz, pts, npoint, 1.2, K, layers)
    assert np.array_equal(idx.cpu().numpy(), ri)
    close(new_points.cpu().numpy(), rp)


def test_sa_module_k16_config0(pn2, oracle, cuda):
    """BASELINE config[0]: B=2, N=1024, npoint=256, K=16, C=3, r=0.2 (K=16 -> unfused path with pool=16)."""
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    rs = np.random.RandomState(7)
    xyz = rs.random_sample((2, 1024, 3)).astype(np.float32)
    pts = rs.random_sample((2, 1024, 3)).astype(np.float32)
    store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=2))
    kw = dict(npoint=256, radius=0.2, nsample=16, mlp=[32, 32, 64], mlp2=None, group_all=False, is_training=False,
              bn_decay=None, scope="cfg0")
    pu.pointnet_sa_module(T(xyz, cuda), T(pts, cuda), **kw)
    randomize_bn(store, 5)
    new_xyz, new_points, idx = pu.pointnet_sa_module(T(xyz, cuda), T(pts, cuda), **kw)
    layers = layer_dicts(store, "cfg0", ["conv0", "conv1", "conv2"])
    r_xyz, r_pts, r_idx = oracle.sa_module(xyz, pts, 256, 0.2, 16, layers)
    assert np.array_equal(idx.cpu().numpy(), r_idx)
    close(new_points.cpu().numpy(), r_pts)


def test_sa_training_path_matches_inference_maths(pn2, oracle, cuda):
    """is_training=True runs the differentiable torch path on the HIP gather ops; with BN statistics
    taken from the batch it must equal a numpy restatement, and gradients must flow to W and points."""
    import torch
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    rs = np.random.RandomState(9)
    xyz = rs.random_sample((2, 256, 3)).astype(np.float32)
    pts = rs.randn(2, 256, 8).astype(np.float32)
    store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=4))
    pt = T(pts, cuda).requires_grad_(True)
    new_xyz, new_points, idx = pu.pointnet_sa_module(T(xyz, cuda), pt, npoint=32, radius=0.4, nsample=32,
                                                     mlp=[32, 64], mlp2=None, group_all=False, is_training=True,
                                                     bn_decay=0.5, scope="tr")
    r_xyz, r_np, r_idx, _ = oracle.sample_and_group(32, 0.4, 32, xyz, pts)
    assert np.array_equal(idx.cpu().numpy(), r_idx)
    h = r_np.astype(np.float64)
    for i in range(2):
        W = store.params["tr/conv%d/weights" % i].detach().cpu().numpy().astype(np.float64)
        W = W.reshape(W.shape[-2], W.shape[-1])
        h = h @ W  # bias 0, gamma 1, beta 0 at init
        mu, var = h.mean((0, 1, 2)), h.var((0, 1, 2))
        h = np.maximum((h - mu) / np.sqrt(var + 1e-3), 0)
    ref = h.max(2)
    got = new_points.detach().cpu().numpy()
    assert np.abs(got - ref).max() < 2e-4  # batch-stat BN in fp32 (torch) vs fp64
    new_points.sum().backward()
    assert pt.grad is not None and float(pt.grad.abs().sum()) > 0
    assert store.params["tr/conv0/weights"].grad is not None
    # moving averages moved away from their init (decay 0.5)
    assert float(store.buffers["tr/conv0/bn/moving_mean"].abs().sum()) > 0


# ------------------------------------------------------------------ FP module ---------------
@pytest.mark.parametrize("n1,n2,c1,c2,mlp", [(256, 64, 128, 256, [256, 256]), (1024, 256, 64, 256, [256, 128]),
                                             (2048, 256, 3, 128, [128, 128, 128]), (300, 40, 0, 32, [64])])
def test_fp_module_inference_vs_oracle(pn2, oracle, cuda, n1, n2, c1, c2, mlp):
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    rs = np.random.RandomState(n1 + c2)
    xyz1 = rs.random_sample((2, n1, 3)).astype(np.float32)
    xyz2 = xyz1[:, rs.permutation(n1)[:n2]].copy()  # sparse level is a subset (exact-zero distances occur)
    p1 = rs.randn(2, n1, c1).astype(np.float32) if c1 else None
    p2 = rs.randn(2, n2, c2).astype(np.float32)
    store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=6))
    args = (T(xyz1, cuda), T(xyz2, cuda), T(p1, cuda) if c1 else None, T(p2, cuda), mlp, False, None)
    pu.pointnet_fp_module(*args, scope="fp")
    randomize_bn(store, 8)
    out = pu.pointnet_fp_module(*args, scope="fp")
    layers = layer_dicts(store, "fp", ["conv_%d" % i for i in range(len(mlp))])
    ref = oracle.fp_module(xyz1, xyz2, p1, p2, layers)
    assert out.shape == (2, n1, mlp[-1])
    close(out.cpu().numpy(), ref)


def test_fp_interp_concat_matches_unfused_ops(pn2, oracle, cuda):
    """fused weights+interpolate+concat == three_interpolate(oracle weights) ++ points1, bit for bit."""
    rs = np.random.RandomState(3)
    xyz1 = rs.random_sample((2, 500, 3)).astype(np.float32)
    xyz2 = xyz1[:, :77].copy()
    p1 = rs.randn(2, 500, 5).astype(np.float32)
    p2 = rs.randn(2, 77, 12).astype(np.float32)
    d, i = pn2.three_nn(T(xyz1, cuda), T(xyz2, cuda))
    got = pn2.util.pointnet_util._fp_interp_concat(d, i, T(p1, cuda), T(p2, cuda)).cpu().numpy()
    rd, ri = oracle.three_nn(xyz1, xyz2)
    ref = np.concatenate([oracle.three_interpolate(p2, ri, oracle.fp_weights(rd)), p1], axis=2)
    assert np.array_equal(got[:, :, 12:], p1)
    assert np.allclose(got, ref, rtol=2e-7, atol=1e-7)  # 1/d and /norm are IEEE on both sides; sum order identical


@pytest.mark.parametrize("n,m,c1,c2,pad_to", [(8192, 256, 3, 128, 8),     # v4 kernel, scalar points1 tail (FP4 layout)
                                                 (1000, 100, 64, 256, 8),    # v4, 16-byte points1 tail (FP3 layout)
                                                 (257, 40, 0, 64, 64),       # v4, no points1, pure zero-pad tail
                                                 (300, 33, 0, 128, 1),       # no tail at all -> row kernel
                                                 (500, 77, 5, 12, 1),        # odd widths -> row kernel
                                                 (64, 16, 256, 512, 8),      # FP1 layout, 12 column chunks
                                                 (100, 9, 7, 1000, 1)])      # > 12 chunks -> one-row kernel
def test_fp_interp_concat_all_kernel_variants(pn2, oracle, cuda, n, m, c1, c2, pad_to):
    """Every dispatch branch of pn2_fp_interp_concat reproduces oracle weights + three_interpolate + concat
    bit for bit (same IEEE divisions, same unfused (p1*w1 + p2*w2) + p3*w3), pad columns are zero."""
    rs = np.random.RandomState(n + c2)
    xyz1 = rs.random_sample((3, n, 3)).astype(np.float32)
    xyz2 = rs.random_sample((3, m, 3)).astype(np.float32)
    xyz2[:, :3] = xyz1[:, :3]  # zero distances -> the 1e-10 clamp
    p1 = rs.randn(3, n, c1).astype(np.float32) if c1 else None
    p2 = rs.randn(3, m, c2).astype(np.float32)
    rd, ri = oracle.three_nn(xyz1, xyz2)
    got = pn2.util.pointnet_util._fp_interp_concat(T(rd, cuda), T(ri, cuda), None if p1 is None else T(p1, cuda),
                                                   T(p2, cuda), pad_to=pad_to).cpu().numpy()
    ref = oracle.three_interpolate(p2, ri, oracle.fp_weights(rd))
    if c1:
        ref = np.concatenate([ref, p1], axis=2)
    cw = -(-(c1 + c2) // pad_to) * pad_to
    assert got.shape == (3, n, cw)
    assert np.array_equal(got[:, :, :c1 + c2], ref)
    assert not got[:, :, c1 + c2:].any()


# ------------------------------------------------------------------ whole stack -------------
def _oracle_stack(oracle, store, pc, hp, pn2):
    xyz, feat = pc[:, :, :3], pc[:, :, 3:6]
    xyzs, feats = [xyz], [feat]
    for li in range(4):
        k = "l%d_" % (li + 1)
        layers = layer_dicts(store, "layer%d" % (li + 1), ["conv%d" % i for i in range(3)])
        nx, npts, _ = oracle.sa_module(xyzs[-1], feats[-1].astype(np.float32), hp[k + "npoint"], hp[k + "radius"],
                                       hp[k + "nsample"], layers)
        xyzs.append(nx)
        feats.append(npts)
    up = feats[4]
    for fi in range(4):
        lvl = 3 - fi
        names = ["conv_%d" % i for i in range(len(pn2.model.FP_MLPS[fi]))]
        layers = layer_dicts(store, "fa_layer%d" % (fi + 1), names)
        up = oracle.fp_module(xyzs[lvl], xyzs[lvl + 1], feats[lvl].astype(np.float32), up.astype(np.float32), layers)
    return up


def test_ssg_stack_small_vs_oracle(pn2, oracle, cuda):
    """The full SA x4 + FP x4 stack (semantic.json radii/nsample, scaled-down npoint) end to end."""
    tfu = pn2.util.tf_util
    hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
    hp.update(l1_npoint=256, l2_npoint=64, l3_npoint=32, l4_npoint=16)
    rs = np.random.RandomState(0)
    pc = np.concatenate([np.asarray(s_scene(1, 2, 2048)), rs.random_sample((2, 2048, 3)).astype(np.float32)], axis=2)
    store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=11))
    pn2.model.get_sa_fp_features(T(pc, cuda), False, hp)
    randomize_bn(store, 12)
    out, _ = pn2.model.get_sa_fp_features(T(pc, cuda), False, hp)
    ref = _oracle_stack(oracle, store, pc, hp, pn2)
    assert out.shape == (2, 2048, 128)
    got = out.cpu().numpy()
    err = np.abs(got - ref)
    # compounded error of 8 chained modules, still inside north_star's 1e-5 (full size: test_fullsize_gpu.py)
    print("small stack end-to-end err %.2e" % (err / (1 + np.abs(ref))).max())
    assert (err <= 1e-5 + 1e-5 * np.abs(ref)).all(), err.max()


def test_get_model_head_shapes(pn2, cuda):
    tfu = pn2.util.tf_util
    hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
    hp.update(l1_npoint=128, l2_npoint=64, l3_npoint=32, l4_npoint=16)
    tfu.set_default_store(tfu.VariableStore(device=cuda, seed=13))
    pc = np.concatenate([s_scene(2, 1, 1024), np.random.RandomState(1).random_sample((1, 1024, 3)).astype(np.float32)], 2)
    logits, ep = pn2.model.get_model(T(pc, cuda), False, 9, hp)
    assert logits.shape == (1, 1024, 9) and ep["feats"].shape == (1, 1024, 128)
    import torch
    assert torch.isfinite(logits).all()


def test_model_head_on_oracle_features_within_1e5(pn2, oracle, cuda):
    """the N1 head alone (fc1 conv1d(128)+BN+ReLU, dropout off, fc2 conv1d(9); model.py:131-146) on the
    ORACLE's SA/FP features -- no compounding with the stack's own 4e-7 hidden in the tolerance -- within north_star's 1e-5
    of (1 + |ref|) against the float64 restatement."""
    tfu = pn2.util.tf_util
    hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
    hp.update(l1_npoint=128, l2_npoint=64, l3_npoint=32, l4_npoint=16)
    store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=21))
    rs = np.random.RandomState(4)
    pc = np.concatenate([s_scene(5, 2, 1024), rs.random_sample((2, 1024, 3)).astype(np.float32)], 2)
    pn2.model.get_model(T(pc, cuda), False, 9, hp)  # creates the variables
    randomize_bn(store, 22)
    feats = _oracle_stack(oracle, store, pc, hp, pn2).astype(np.float32)  # what the reference's stack hands to its head
    (fc1,) = layer_dicts(store, None, ["fc1"])
    (fc2,) = layer_dicts(store, None, ["fc2"], bn=False)
    ref = oracle.model_head(feats.astype(np.float64), fc1, fc2)
    import torch
    with torch.no_grad():
        logits = pn2.model.get_head(T(feats, cuda), False, 9)
    err = np.abs(logits.cpu().numpy().astype(np.float64) - ref) / (1.0 + np.abs(ref))
    print("head on oracle features: max err %.2e of (1+|ref|)" % err.max())
    assert err.max() <= 1e-5, err.max()


def test_get_model_logits_and_loss_vs_oracle(pn2, oracle, cuda):
    """the head (fc1 conv1d+BN+ReLU, dropout off, fc2) on top of the SA/FP stack and the weighted
    sparse cross-entropy (SUM_BY_NONZERO_WEIGHTS) against their numpy restatements."""
    import torch
    tfu = pn2.util.tf_util
    hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
    hp.update(l1_npoint=128, l2_npoint=64, l3_npoint=32, l4_npoint=16)
    store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=21))
    rs = np.random.RandomState(4)
    pc = np.concatenate([s_scene(5, 2, 1024), rs.random_sample((2, 1024, 3)).astype(np.float32)], 2)
    pn2.model.get_model(T(pc, cuda), False, 9, hp)  # creates the variables
    randomize_bn(store, 22)
    logits, ep = pn2.model.get_model(T(pc, cuda), False, 9, hp)
    feats = _oracle_stack(oracle, store, pc, hp, pn2)
    (fc1,) = layer_dicts(store, None, ["fc1"])
    (fc2,) = layer_dicts(store, None, ["fc2"], bn=False)
    ref = oracle.model_head(feats, fc1, fc2)
    got = logits.cpu().numpy().astype(np.float64)
    err = np.abs(got - ref)
    assert (err <= 5e-5 + 5e-5 * np.abs(ref)).all(), err.max()
    labels = rs.randint(0, 9, (2, 1024))
    smpw = (rs.random_sample((2, 1024)) * 2).astype(np.float32)
    smpw[0, :100] = 0.0  # zero-weight points do not count in the denominator
    loss = pn2.model.get_loss(logits, T(labels.astype(np.int64), cuda), T(smpw, cuda))
    ref_loss = oracle.weighted_sparse_ce(got, labels, smpw)
    assert abs(float(loss) - ref_loss) < 1e-5 * max(1.0, abs(ref_loss))
    zero = pn2.model.get_loss(logits, T(labels.astype(np.int64), cuda), torch.zeros(2, 1024, device=cuda))
    assert float(zero) == 0.0 == oracle.weighted_sparse_ce(got, labels, np.zeros((2, 1024)))


def test_graph_forward_matches_eager(pn2, cuda):
    """The hipGraph replay of the forward pass (runtime.CapturedForward, what bench.py times) is bit-identical to the eager
    forward (same kernels, same inputs), also on a new input pushed through the static buffer."""
    import torch
    tfu = pn2.util.tf_util
    hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
    hp.update(l1_npoint=256, l2_npoint=64, l3_npoint=32, l4_npoint=16)
    rs = np.random.RandomState(3)
    pc = np.concatenate([s_scene(4, 2, 2048), rs.random_sample((2, 2048, 3)).astype(np.float32)], axis=2)
    store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=21))
    x = T(pc, cuda)
    with torch.no_grad():
        ref, _ = pn2.model.get_sa_fp_features(x, False, hp)
        randomize_bn(store, 22)
        ref, _ = pn2.model.get_sa_fp_features(x, False, hp)
    torch.cuda.synchronize()
    cap = pn2.runtime.CapturedForward(lambda t: pn2.model.get_sa_fp_features(t, False, hp)[0], x)
    for _ in range(3):
        out = cap.replay()
    torch.cuda.synchronize()
    assert torch.equal(ref, out)
    x2 = T(np.ascontiguousarray(pc[::-1]), cuda)  # new input through the static buffer
    out2 = cap(x2).clone()
    with torch.no_grad():
        ref2, _ = pn2.model.get_sa_fp_features(x2, False, hp)
    torch.cuda.synchronize()
    assert torch.equal(ref2, out2)


# ------------------------------------------------------------------ pn2_mlp_chain -----------
@pytest.mark.parametrize("rows,cin,widths,pool", [(4096, 136, [128, 128], 0), (1000, 128, [128], 0), (777, 131, [128, 128], 0),
                                                  (2048, 64, [64, 128], 0), (3200, 128, [128, 128], 32), (96, 40, [64], 32)])
def test_mlp_chain_vs_fp64(pn2, cuda, rows, cin, widths, pool):
    rs = np.random.RandomState(rows + cin)
    x = rs.randn(rows, cin).astype(np.float32)
    ws, bs, ref = [], [], x.astype(np.float64)
    c = cin
    for w_ in widths:
        W = (rs.randn(c, w_) / np.sqrt(c)).astype(np.float32)
        b = (rs.randn(w_) * 0.1).astype(np.float32)
        ws.append(T(W, cuda)); bs.append(T(b, cuda))
        ref = np.maximum(ref @ W.astype(np.float64) + b, 0)
        c = w_
    if pool:
        ref = ref.reshape(rows // pool, pool, widths[-1]).max(1)
    y = pn2.util.tf_util.hip_mlp_chain(T(x, cuda), ws, bs, pool=pool)
    assert y is not None
    close(y.cpu().numpy(), ref)


def test_mlp_chain_rejects_what_does_not_fit(pn2, cuda):
    import torch
    x = torch.zeros(64, 128, device=cuda)
    w = torch.zeros(128, 256, device=cuda)
    b = torch.zeros(256, device=cuda)
    assert pn2.util.tf_util.hip_mlp_chain(x, [w], [b]) is None  # width 256 > 128 -> caller falls back


@pytest.mark.parametrize("mlp,c,m", [([64, 64, 128], 64, 256), ([128, 128, 256], 128, 64), ([128, 128, 128], 32, 50)])
def test_sa_module_hoisted_first_layer_equals_in_place(pn2, oracle, cuda, mlp, c, m):
    """pn2_sa_mlp_fused_pre: the FEATURE part of the first SA layer computed once per source point (zf = points @ W1[3:],
    n rows instead of m*K grouped rows) and gathered into the accumulators, the xyz part still on the MFMA.  Same products in
    a different summation order: equals the in-place kernel to fp32 rounding and the fp64 oracle at 1e-5 (SA2 / SA3 shapes of
    semantic.json + an odd one)."""
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    rs = np.random.RandomState(m)
    xyz = s_scene(m, 2, 1024)
    pts = rs.randn(2, 1024, c).astype(np.float32)
    store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=51))
    kw = dict(npoint=m, radius=1.0, nsample=32, mlp=mlp, mlp2=None, group_all=False, is_training=False, bn_decay=None, scope="sa")
    pu.pointnet_sa_module(T(xyz, cuda), T(pts, cuda), **kw)
    randomize_bn(store, 52)
    outs = {}
    for hoist in (True, False):
        calls = []
        pn2._lib.lib.trace = calls
        pu.USE_HOISTED_SA = hoist
        try:
            _, outs[hoist], idx = pu.pointnet_sa_module(T(xyz, cuda), T(pts, cuda), **kw)
        finally:
            pn2._lib.lib.trace = None
            pu.USE_HOISTED_SA = True
        assert ("pn2_sa_mlp_fused_pre" in [c_[0] for c_ in calls]) == hoist
    a, b_ = outs[True].cpu().numpy(), outs[False].cpu().numpy()
    np.testing.assert_allclose(a, b_, rtol=1e-5, atol=1e-5)
    _, ref, ridx = oracle.sa_module(xyz, pts, m, 1.0, 32, layer_dicts(store, "sa", ["conv0", "conv1", "conv2"]))
    assert np.array_equal(idx.cpu().numpy(), ridx)
    close(a, ref)
    close(b_, ref)


@pytest.mark.parametrize("kind,mlp,c,c1", [("sa", [64, 64, 128], 64, 0), ("sa", [128], 32, 0), ("sa", [256, 512], 256, 0),
                                           ("fp", [128, 128, 128], 128, 3), ("fp", [64], 32, 3)])
def test_training_first_layer_on_the_source_rows_equals_the_grouped_form(pn2, cuda, kind, mlp, c, c1):
    """Training path: the feature half of a module's first conv applied to the SOURCE rows (tf_util._TrainHoistedBnRelu,
    pn2_sa_hoist_rows / pn2_fp_hoist_rows; the grouped / concatenated tensor is never built, GEMM + data + weight gradient on
    n resp. m rows) against the grouped form (pn2_sa_group_concat / pn2_fp_interp_concat + conv2d): module output, moving
    averages, gradient w.r.t. the source features and every parameter gradient.  Reference maths:
    pointnet_util.py:39-54,150-170 (SA) and :300-325 (FP) with batch-statistics BN (tf_util.py:555-581)."""
    import torch
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    rs = np.random.RandomState(len(mlp) + c)
    b, n = 2, 1024
    xyz = T(s_scene(c, b, n), cuda)
    if kind == "sa":
        m, ns = 128, 16
        src0 = T(rs.randn(b, n, c).astype(np.float32), cuda)
        new_xyz, idx = pu.sa_geometry(xyz, m, 0.8, ns)
        plan = pu.scatter_plan(idx, n)
        oshape = (b, m, mlp[-1])
    else:
        m = 128
        xyz2 = xyz[:, :m].contiguous()
        src0 = T(rs.randn(b, m, c).astype(np.float32), cuda)
        p1_0 = T(rs.rand(b, n, c1).astype(np.float32), cuda)
        dist, idx = pn2.three_nn(xyz, xyz2)
        plan = pu.scatter_plan(idx, m, dist, weight_kind=2)
        oshape = (b, n, mlp[-1])
    probe = torch.sin(torch.arange(int(np.prod(oshape)), device=cuda).float() * 0.11).reshape(oshape)
    outs = {}
    for hoist in (True, False):
        tfu.USE_HOISTED_TRAIN = hoist
        store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=9))
        calls = []
        pn2._lib.lib.trace = calls
        try:
            tfu.reset_bn_links()
            src = src0.clone().requires_grad_(True)
            if kind == "sa":
                _, out, _ = pu.pointnet_sa_module(xyz, src, m, 0.8, ns, mlp, None, False, True, 0.5, "mod",
                                                  geometry=(new_xyz, idx, plan))
            else:
                p1 = p1_0.clone()  # the colours of the level-0 module: no gradient
                out = pu.pointnet_fp_module(xyz, xyz2, p1, src, mlp, True, 0.5, "mod", nn=(dist, idx, plan))
            assert tuple(out.shape) == oshape
            (out * probe).sum().backward()
        finally:
            pn2._lib.lib.trace = None
            tfu.USE_HOISTED_TRAIN = True
        names = [c_[0] for c_ in calls]
        hname = "pn2_sa_hoist_rows" if kind == "sa" else "pn2_fp_hoist_rows"   # (+ "_bn": the layer's statistics taken on the way out)
        assert ((hname in names) or (hname + "_bn" in names)) == hoist
        assert (("pn2_sa_group_concat" if kind == "sa" else "pn2_fp_interp_concat") in names) == (not hoist)
        outs[hoist] = [out.detach(), src.grad] + [p_.grad for _, p_ in sorted(store.params.items()) if p_.grad is not None] + \
                      [v.clone() for _, v in sorted(store.buffers.items())]
    assert len(outs[True]) == len(outs[False]) and len(outs[True]) >= 4
    for a, r in zip(outs[True], outs[False]):
        assert a.shape == r.shape
        sc = max(float(r.abs().max()), 1e-3)
        assert float((a - r).abs().max()) <= 3e-4 * sc, (a.shape, float((a - r).abs().max()), sc)


@pytest.mark.parametrize("kind,mlp,c,plan_on", [("sa", [32, 32, 64], 3, False), ("sa", [64, 64, 128], 64, True),
                                                ("fp", [256, 128], 64, False), ("fp", [128, 128, 128], 128, True)])
def test_training_stack_with_batch_norm_applied_on_load_equals_the_materialised_form(pn2, cuda, kind, mlp, c, plan_on):
    """Training path: inside an MLP stack a layer publishes its batch-norm (scale, shift) and hands over the UN-normalised
    output (pn2_bn_relu_forward_deferred); the next layer's forward GEMM and weight gradient apply relu(fma(y, scale, shift))
    while loading it (pn2_linear_bn_stats_xf / pn2_linear_wgrad_accumulate_xf).  Against the form that writes the normalised
    activation (tf_util.USE_BN_ON_LOAD = False): module output, moving averages, source gradient, every parameter
    gradient -- for plain stacks and for stacks whose first layer runs on the source rows.  tf_util.py:186-204,555-581."""
    import torch
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    rs = np.random.RandomState(len(mlp) + c)
    b, n = 3, 1024
    xyz = T(s_scene(c + 1, b, n), cuda)
    if kind == "sa":
        m, ns = 128, 16
        src0 = T(rs.randn(b, n, c).astype(np.float32), cuda)
        new_xyz, idx = pu.sa_geometry(xyz, m, 0.8, ns)
        geo = (new_xyz, idx, pu.scatter_plan(idx, n)) if plan_on else (new_xyz, idx)
        oshape = (b, m, mlp[-1])
    else:
        m = 128
        xyz2 = xyz[:, :m].contiguous()
        src0 = T(rs.randn(b, m, 96).astype(np.float32), cuda)
        p1 = T(rs.rand(b, n, 3 if plan_on else c).astype(np.float32), cuda)
        dist, idx = pn2.three_nn(xyz, xyz2)
        nn = (dist, idx, pu.scatter_plan(idx, m, dist, weight_kind=2)) if plan_on else (dist, idx)
        oshape = (b, n, mlp[-1])
    probe = torch.sin(torch.arange(int(np.prod(oshape)), device=cuda).float() * 0.13).reshape(oshape)
    outs = {}
    for on in (True, False):
        tfu.USE_BN_ON_LOAD = on
        tfu.USE_BN_GRAD_ON_LOAD = False  # this test pins the forward transform + its weight gradient; the gradient side has its own
        tfu.USE_BN_FINISH_IN_PRODUCER = False  # ... and the separate-launch form of the entry points it counts
        store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=11))
        calls = []
        pn2._lib.lib.trace = calls
        try:
            tfu.reset_bn_links()
            src = src0.clone().requires_grad_(True)
            if kind == "sa":
                _, out, _ = pu.pointnet_sa_module(xyz, src, m, 0.8, ns, mlp, None, False, True, 0.5, "mod", geometry=geo)
            else:
                out = pu.pointnet_fp_module(xyz, xyz2, p1, src, mlp, True, 0.5, "mod", nn=nn)
            (out * probe).sum().backward()
        finally:
            pn2._lib.lib.trace = None
            tfu.USE_BN_ON_LOAD = True
            tfu.USE_BN_GRAD_ON_LOAD = True
            tfu.USE_BN_FINISH_IN_PRODUCER = True
        names = [c_[0] for c_ in calls]
        assert names.count("pn2_bn_relu_forward_deferred") == (len(mlp) - 1 if on else 0), names
        assert names.count("pn2_linear_bn_stats_xf") == (len(mlp) - 1 if on else 0)
        assert names.count("pn2_linear_wgrad_accumulate_xf") == (len(mlp) - 1 if on else 0)
        outs[on] = [out.detach(), src.grad] + [p_.grad for _, p_ in sorted(store.params.items()) if p_.grad is not None] + \
                   [v.clone() for _, v in sorted(store.buffers.items())]
    assert len(outs[True]) == len(outs[False]) and len(outs[True]) >= 6
    for a, r in zip(outs[True], outs[False]):
        sc = max(float(r.abs().max()), 1e-3)
        assert float((a - r).abs().max()) <= 2e-4 * sc, (a.shape, float((a - r).abs().max()), sc)


@pytest.mark.parametrize("kind,mlp,c,plan_on,ns", [("sa", [32, 32, 64], 3, False, 32), ("sa", [64, 64, 128], 64, True, 32),
                                                   ("sa", [128, 128, 256], 128, True, 32), ("sa", [64, 128], 16, False, 16),
                                                   ("fp", [256, 128], 64, False, 0), ("fp", [128, 128, 128], 128, True, 0),
                                                   ("fp", [256, 256], 512, False, 0)])
def test_bn_grad_on_load_equals_the_materialised_form(pn2, cuda, kind, mlp, c, plan_on, ns):
    """Round 6: the gradient LEAVING a layer's batch norm is formed by the layer's data and weight gradient GEMMs while they load
    (y, dz) (pn2_bn_grad_constants -> pn2_linear_dgrad_gx / pn2_linear_wgrad_gx), also behind the fused max over 32 neighbours,
    instead of being written by pn2_bn_relu_backward and re-read twice.  Both forms use the same float expressions: the source
    gradient, every parameter gradient and the module output must agree to the rounding of the (atomic) reduction orders.
    Reference maths: util/tf_util.py:555-581 + :181-186 through tf.gradients."""
    import torch
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    rs = np.random.RandomState(len(mlp) + c)
    b, n = 3, 1024
    xyz = T(s_scene(c + 1, b, n), cuda)
    if kind == "sa":
        m = 128
        src0 = T(rs.randn(b, n, c).astype(np.float32), cuda)
        new_xyz, idx = pu.sa_geometry(xyz, m, 0.8, ns)
        geo = (new_xyz, idx, pu.scatter_plan(idx, n)) if plan_on else (new_xyz, idx)
        oshape = (b, m, mlp[-1])
    else:
        m = 128
        xyz2 = xyz[:, :m].contiguous()
        src0 = T(rs.randn(b, m, 96).astype(np.float32), cuda)
        p1 = T(rs.rand(b, n, 3 if plan_on else c).astype(np.float32), cuda)
        dist, idx = pn2.three_nn(xyz, xyz2)
        nn = (dist, idx, pu.scatter_plan(idx, m, dist, weight_kind=2)) if plan_on else (dist, idx)
        oshape = (b, n, mlp[-1])
    probe = torch.sin(torch.arange(int(np.prod(oshape)), device=cuda).float() * 0.13).reshape(oshape)
    outs, counts = {}, {}
    for on in (True, False):
        tfu.USE_BN_GRAD_ON_LOAD = on
        store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=11))
        calls = []
        pn2._lib.lib.trace = calls
        try:
            tfu.reset_bn_links()
            src = src0.clone().requires_grad_(True)
            if kind == "sa":
                _, out, _ = pu.pointnet_sa_module(xyz, src, m, 0.8, ns, mlp, None, False, True, 0.5, "mod", geometry=geo)
            else:
                out = pu.pointnet_fp_module(xyz, xyz2, p1, src, mlp, True, 0.5, "mod", nn=nn)
            (out * probe).sum().backward()
        finally:
            pn2._lib.lib.trace = None
            tfu.USE_BN_GRAD_ON_LOAD = True
        names = [c_[0] for c_ in calls]
        counts[on] = {k: names.count(k) for k in ("pn2_bn_grad_constants", "pn2_linear_dgrad_gx", "pn2_linear_wgrad_gx",
                                                  "pn2_linear_bwd_fused")}
        outs[on] = [out.detach(), src.grad] + [p_.grad for _, p_ in sorted(store.params.items()) if p_.grad is not None]
    # every layer but a source-row first layer (whose dy feeds the scatter plan) takes the on-load form; a 16-neighbour pool does not
    hoisted = 1 if plan_on else 0
    pooled_other = 1 if (kind == "sa" and ns != 32) else 0
    want = len(mlp) - hoisted - pooled_other
    # (a layer whose dz is the data gradient of the layer above gets its constants from that GEMM's last workgroup: only the top
    # layer of the stack still calls pn2_bn_grad_constants; the data gradients run through pn2_linear_dgrad_fin)
    # (32 / 64-channel layers that need both gradients take them from ONE launch: pn2_linear_bwd_fused)
    assert counts[True]["pn2_linear_wgrad_gx"] + counts[True]["pn2_linear_bwd_fused"] == want, counts
    assert counts[True]["pn2_bn_grad_constants"] <= min(want, 1), counts
    narrow = sum(1 for i in range(1, len(mlp)) if mlp[i - 1] in (32, 64) and mlp[i] in (32, 64) and not (i == len(mlp) - 1 and pooled_other))
    assert counts[True]["pn2_linear_bwd_fused"] == narrow, (counts, narrow)
    assert not any(counts[False].values()), counts
    assert len(outs[True]) == len(outs[False]) and len(outs[True]) >= 6
    for a, r in zip(outs[True], outs[False]):
        sc = max(float(r.abs().max()), 1e-3)
        assert float((a - r).abs().max()) <= 1e-5 * sc, (a.shape, float((a - r).abs().max()), sc)


@pytest.mark.parametrize("kind,mlp,c,plan_on,ns", [("sa", [32, 32, 64], 3, False, 32), ("sa", [64, 64, 128], 64, True, 32),
                                                   ("sa", [256, 256, 512], 256, True, 32), ("fp", [256, 128], 64, False, 0),
                                                   ("fp", [128, 128, 128], 128, True, 0)])
def test_bn_finish_in_the_producer_equals_the_separate_launches(pn2, cuda, kind, mlp, c, plan_on, ns):
    """Round 6: the one-block launches that followed every producer of batch-norm sums (fold the slot copies, derive scale / shift
    resp. the gradient constants) are done by the producer's last workgroup (two-level ticket, pn2_common.h pn2_bn_finish):
    module output, moving averages, source gradient and every parameter gradient against the separate launches
    (tf_util.USE_BN_FINISH_IN_PRODUCER = False), and the launches are really gone.  tf_util.py:186-204,555-581."""
    import torch
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    rs = np.random.RandomState(len(mlp) + c)
    b, n = 3, 1024
    xyz = T(s_scene(c + 1, b, n), cuda)
    if kind == "sa":
        m = 128
        src0 = T(rs.randn(b, n, c).astype(np.float32), cuda)
        new_xyz, idx = pu.sa_geometry(xyz, m, 0.8, ns)
        geo = (new_xyz, idx, pu.scatter_plan(idx, n)) if plan_on else (new_xyz, idx)
        oshape = (b, m, mlp[-1])
    else:
        m = 128
        xyz2 = xyz[:, :m].contiguous()
        src0 = T(rs.randn(b, m, 96).astype(np.float32), cuda)
        p1 = T(rs.rand(b, n, 3 if plan_on else c).astype(np.float32), cuda)
        dist, idx = pn2.three_nn(xyz, xyz2)
        nn = (dist, idx, pu.scatter_plan(idx, m, dist, weight_kind=2)) if plan_on else (dist, idx)
        oshape = (b, n, mlp[-1])
    probe = torch.sin(torch.arange(int(np.prod(oshape)), device=cuda).float() * 0.13).reshape(oshape)
    outs, names = {}, {}
    for on in (True, False):
        tfu.USE_BN_FINISH_IN_PRODUCER = on
        store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=11))
        calls = []
        pn2._lib.lib.trace = calls
        try:
            tfu.reset_bn_links()
            src = src0.clone().requires_grad_(True)
            if kind == "sa":
                _, out, _ = pu.pointnet_sa_module(xyz, src, m, 0.8, ns, mlp, None, False, True, 0.5, "mod", geometry=geo)
            else:
                out = pu.pointnet_fp_module(xyz, xyz2, p1, src, mlp, True, 0.5, "mod", nn=nn)
            (out * probe).sum().backward()
        finally:
            pn2._lib.lib.trace = None
            tfu.USE_BN_FINISH_IN_PRODUCER = True
        names[on] = [c_[0] for c_ in calls]
        outs[on] = [out.detach(), src.grad] + [p_.grad for _, p_ in sorted(store.params.items()) if p_.grad is not None] + \
                   [v.clone() for _, v in sorted(store.buffers.items())]
    assert "pn2_linear_bn_stats_fin" in names[True] and "pn2_linear_dgrad_fin" in names[True]
    assert "pn2_linear_bn_stats_fin" not in names[False] and "pn2_linear_dgrad_fin" not in names[False]
    assert "pn2_linear_bn_stats_xf" not in names[True] and "pn2_linear_bn_stats" not in names[True]
    assert len(names[True]) < len(names[False])
    assert len(outs[True]) == len(outs[False]) and len(outs[True]) >= 6
    for a, r in zip(outs[True], outs[False]):
        sc = max(float(r.abs().max()), 1e-3)
        assert float((a - r).abs().max()) <= 1e-5 * sc, (a.shape, float((a - r).abs().max()), sc)


@pytest.mark.parametrize("c,mlp,ns,with_geo", [(3, [32, 32, 64], 32, True), (3, [64], 32, False), (5, [48, 96], 16, True),
                                               (1, [128, 128], 32, True)])
def test_sa_first_layer_in_one_launch_equals_the_separate_ops(pn2, cuda, c, mlp, ns, with_geo):
    """Round 6: the first layer of an SA module with few point channels (the level-0 module: xyz + rgb) -- gather, centre, concat,
    the (3 + c) -> w product, its batch statistics and their fold / constants -- as ONE launch (pn2_sa_first_layer_bn) against
    pn2_sa_group_concat + the MFMA layer: module output, moving averages, every parameter gradient.
    pointnet_util.py:39-54,150-156, tf_util.py:181-204."""
    import torch
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    rs = np.random.RandomState(c + len(mlp))
    b, n, m = 3, 1024, 128
    xyz = T(s_scene(c + 1, b, n), cuda)
    pts = T(rs.rand(b, n, c).astype(np.float32), cuda)
    geo = pu.sa_geometry(xyz, m, 0.8, ns) if with_geo else None
    probe = torch.sin(torch.arange(b * m * mlp[-1], device=cuda).float() * 0.13).reshape(b, m, mlp[-1])
    outs, names = {}, {}
    for on in (True, False):
        tfu.USE_SA_FIRST_LAYER_FUSED = on
        store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=11))
        calls = []
        pn2._lib.lib.trace = calls
        try:
            tfu.reset_bn_links()
            new_xyz, out, idx = pu.pointnet_sa_module(xyz, pts, m, 0.8, ns, mlp, None, False, True, 0.5, "mod", geometry=geo)
            (out * probe).sum().backward()
        finally:
            pn2._lib.lib.trace = None
            tfu.USE_SA_FIRST_LAYER_FUSED = True
        names[on] = [c_[0] for c_ in calls]
        outs[on] = [out.detach(), new_xyz, idx.float()] + [p_.grad for _, p_ in sorted(store.params.items()) if p_.grad is not None] + \
                   [v.clone() for _, v in sorted(store.buffers.items())]
    assert names[True].count("pn2_sa_first_layer_bn") == 1 and "pn2_sa_group_concat" not in names[True]
    assert "pn2_sa_first_layer_bn" not in names[False] and "pn2_sa_group_concat" in names[False]
    assert len(outs[True]) == len(outs[False]) and len(outs[True]) >= 7
    for a, r in zip(outs[True], outs[False]):
        sc = max(float(r.abs().max()), 1e-3)
        assert float((a - r).abs().max()) <= 2e-5 * sc, (a.shape, float((a - r).abs().max()), sc)


@pytest.mark.parametrize("kind,b,n,m,ns,c1,cout", [
    ("sa", 16, 1024, 256, 32, 0, 64),     # SA2 of configs[1]: 131072 rows
    ("sa", 3, 500, 37, 16, 0, 132),       # ragged: rows per cloud not a multiple of the row slots, cout / 4 = 33 columns
    ("sa", 2, 256, 64, 8, 0, 1024),       # the widest row the kernel takes: one row slot per workgroup
    ("fp", 16, 8192, 1024, 0, 3, 128),    # FP4 of configs[1]
    ("fp", 2, 1000, 100, 0, 5, 36),
])
def test_hoist_rows_take_the_batch_statistics_on_the_way_out(pn2, cuda, kind, b, n, m, ns, c1, cout):
    """Round 6: pn2_sa_hoist_rows_bn / pn2_fp_hoist_rows_bn -- the hoisted first layer of an SA / FP module (csrc/pn2_hoist.hip) that
    also leaves the batch statistics of what it writes, folds them and publishes the deferred batch norm's constants (finish 2) --
    against the plain entry point (y bit for bit) and float64 moments of that y (tf_util.py:186-204,555-581: batch_norm_template
    with is_training=True; the moving averages as tf's fused batch norm keeps them).  Three times in a row: the ticket must elect
    exactly one finishing workgroup each time."""
    import torch
    from ctypes import c_float
    L, P = pn2._lib.lib, pn2._lib.ptr
    rs = np.random.RandomState(b + n + cout)
    st = pn2._lib.stream_ptr()
    if kind == "sa":
        xyz = T(s_scene(3, b, n), cuda)
        new_xyz, idx = pn2.util.pointnet_util.sa_geometry(xyz, m, 0.9, ns)
        z = T(rs.randn(b, n, cout).astype(np.float32), cuda)
        wa = T(rs.randn(3, cout).astype(np.float32), cuda)
        rows = b * m * ns
        plain = lambda y, a: L.pn2_sa_hoist_rows(b, n, m, ns, cout, P(xyz), P(new_xyz), P(idx), P(z), P(wa), P(y), P(a), st)  # noqa: E731
        fused = lambda y, a, *bn: L.pn2_sa_hoist_rows_bn(b, n, m, ns, cout, P(xyz), P(new_xyz), P(idx), P(z), P(wa), P(y), P(a), *bn, st)  # noqa: E731
        gx = lambda: torch.empty(rows, 3, device=cuda)  # noqa: E731
    else:
        xyz = T(s_scene(4, b, n), cuda)
        dist, idx = pn2.three_nn(xyz, xyz[:, :m].contiguous())
        z = T(rs.randn(b, m, cout).astype(np.float32), cuda)
        p1 = T(rs.rand(b, n, c1).astype(np.float32), cuda)
        wa = T(rs.randn(c1, cout).astype(np.float32), cuda)
        rows = b * n
        plain = lambda y, a: L.pn2_fp_hoist_rows(b, n, m, c1, cout, P(dist), P(idx), P(p1), P(z), P(wa), P(y), st)  # noqa: E731
        fused = lambda y, a, *bn: L.pn2_fp_hoist_rows_bn(b, n, m, c1, cout, P(dist), P(idx), P(p1), P(z), P(wa), P(y), *bn, st)  # noqa: E731
        gx = lambda: None  # noqa: E731
    y0, a0 = torch.empty(rows, cout, device=cuda), gx()
    assert plain(y0, a0) == 0
    gamma = T((1.0 + 0.1 * rs.randn(cout)).astype(np.float32), cuda)
    beta, bias = T((0.1 * rs.randn(cout)).astype(np.float32), cuda), T((0.2 * rs.randn(cout)).astype(np.float32), cuda)
    yd = y0.double()
    mean, var = yd.mean(0), yd.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-3)
    nbytes = L.pn2_bn_workspace_bytes(cout)
    for it in range(3):
        ws = torch.zeros(nbytes // 8, dtype=torch.float64, device=cuda)
        rm, rv = torch.zeros(cout, device=cuda), torch.ones(cout, device=cuda)
        sm, si, sc, sh = (torch.empty(cout, device=cuda) for _ in range(4))
        y1, a1 = torch.empty(rows, cout, device=cuda), gx()
        assert fused(y1, a1, P(ws), nbytes, 2, P(gamma), P(beta), P(bias), c_float(1e-3), c_float(0.5), P(rm), P(rv), P(sm), P(si),
                     P(sc), P(sh)) == 0
        assert torch.equal(y1, y0) and (a0 is None or torch.equal(a1, a0)), it
        assert float((sm.double() - mean).abs().max()) <= 1e-5 * max(1.0, float(mean.abs().max())), it
        assert float(((si.double() - invstd) / invstd).abs().max()) <= 1e-5, it
        assert float((sc.double() - gamma.double() * invstd).abs().max()) <= 1e-5 * float((gamma.double() * invstd).abs().max()), it
        assert float((sh.double() - (beta.double() - mean * gamma.double() * invstd)).abs().max()) <= 1e-4, it
        var_unb = var * (rows / (rows - 1.0))
        assert float((rm.double() - 0.5 * (mean + bias.double())).abs().max()) <= 1e-5 * max(1.0, float(mean.abs().max())), it
        assert float((rv.double() - (0.5 + 0.5 * var_unb)).abs().max()) <= 1e-5 * max(1.0, float(var.max())), it
    # refused before any launch
    assert L.pn2_bn_workspace_bytes(cout) > 0
    assert fused(y1, a1, None, nbytes, 2, P(gamma), P(beta), None, c_float(1e-3), c_float(0.5), None, None, P(sm), P(si), P(sc), P(sh)) == -2
    assert fused(y1, a1, P(ws), 8, 2, P(gamma), P(beta), None, c_float(1e-3), c_float(0.5), None, None, P(sm), P(si), P(sc), P(sh)) == -1
    assert fused(y1, a1, P(ws), nbytes, 7, P(gamma), P(beta), None, c_float(1e-3), c_float(0.5), None, None, P(sm), P(si), P(sc), P(sh)) == -1


@pytest.mark.parametrize("kind", ["sa", "fp"])
def test_hoisted_layer_with_fused_statistics_equals_the_two_launches(pn2, cuda, kind):
    """... and at module level: tf_util.USE_HOIST_BN_STATS on / off -- output, moving averages, every gradient of an SA / FP module in
    training mode agree to summation order; on = no statistics kernel behind the hoisted layer."""
    import torch
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    rs = np.random.RandomState(5)
    b, n, c, mlp = 2, (1024 if kind == "sa" else 4096), 32, [64, 64, 128]   # (> 2048 rows: the layer defers its batch norm)
    xyz = T(s_scene(9, b, n), cuda)
    if kind == "sa":
        m, ns = 128, 32
        src0 = T(rs.randn(b, n, c).astype(np.float32), cuda)
        new_xyz, idx = pu.sa_geometry(xyz, m, 0.8, ns)
        plan = pu.scatter_plan(idx, n)
        oshape = (b, m, mlp[-1])
    else:
        m = 128
        xyz2 = xyz[:, :m].contiguous()
        src0 = T(rs.randn(b, m, c).astype(np.float32), cuda)
        p1_0 = T(rs.rand(b, n, 3).astype(np.float32), cuda)
        dist, idx = pn2.three_nn(xyz, xyz2)
        plan = pu.scatter_plan(idx, m, dist, weight_kind=2)
        oshape = (b, n, mlp[-1])
    probe = torch.sin(torch.arange(int(np.prod(oshape)), device=cuda).float() * 0.11).reshape(oshape)
    outs, names = {}, {}
    for on in (True, False):
        tfu.USE_HOIST_BN_STATS, min_rows, tfu.HOIST_BN_STATS_MIN_ROWS = on, tfu.HOIST_BN_STATS_MIN_ROWS, 0
        store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=9))
        calls = []
        pn2._lib.lib.trace = calls
        try:
            tfu.reset_bn_links()
            src = src0.clone().requires_grad_(True)
            if kind == "sa":
                _, out, _ = pu.pointnet_sa_module(xyz, src, m, 0.8, ns, mlp, None, False, True, 0.5, "mod", geometry=(new_xyz, idx, plan))
            else:
                out = pu.pointnet_fp_module(xyz, xyz2, p1_0.clone(), src, mlp, True, 0.5, "mod", nn=(dist, idx, plan))
            (out * probe).sum().backward()
        finally:
            pn2._lib.lib.trace = None
            tfu.USE_HOIST_BN_STATS, tfu.HOIST_BN_STATS_MIN_ROWS = True, min_rows
        names[on] = [c_[0] for c_ in calls]
        outs[on] = [out.detach(), src.grad] + [p_.grad for _, p_ in sorted(store.params.items()) if p_.grad is not None] + \
                   [v.clone() for _, v in sorted(store.buffers.items())]
    hname = "pn2_sa_hoist_rows" if kind == "sa" else "pn2_fp_hoist_rows"
    assert hname + "_bn" in names[True] and hname not in names[True]
    assert hname in names[False] and hname + "_bn" not in names[False]
    assert names[True].count("pn2_bn_relu_forward_deferred") == names[False].count("pn2_bn_relu_forward_deferred") - 1
    for a, r in zip(outs[True], outs[False]):
        sc = max(float(r.abs().max()), 1e-3)
        assert float((a - r).abs().max()) <= 2e-5 * sc, (a.shape, float((a - r).abs().max()), sc)


@pytest.mark.parametrize("rows,cin,cout", [(65536, 32, 32), (524288, 32, 64), (131072, 64, 64), (131072, 64, 128), (65536, 32, 128),
                                           (98304, 64, 32), (131072, 128, 128), (65568, 128, 128)])
@pytest.mark.parametrize("xf", [False, True])
def test_streaming_forward_of_the_narrow_layers_equals_the_tiled_kernel(pn2, cuda, rows, cin, cout, xf):
    """Round 6: fwd_narrow_kernel (csrc/pn2_fwd_narrow.h: 32 / 64 input channels, 32 / 64 / 128 outputs, >= 65536 rows, rows % 32 == 0)
    and fwd_wide_in_kernel (128 -> 128: eight waves per workgroup, the operand tile staged in two slices of 64 channels; 65568 rows = a
    tile count that is not a multiple of the eight waves) behind pn2_linear_bn_stats_fin, without / with the batch norm of the layer below applied on load.  y is the tiled linear_kernel's
    bit for bit -- reached through the same entry point on rows + 1 rows, a shape the streaming kernel refuses -- and the published
    moments / constants are those of float64 sums over that y; three times in a row (the ticket).  tf_util.py:181-204,555-581."""
    import torch
    tfu = pn2.util.tf_util
    rs = np.random.RandomState(rows % 977 + cin + cout)
    x1 = T(rs.randn(rows + 1, cin).astype(np.float32), cuda)
    x = x1[:rows]
    w = T((rs.randn(cin, cout) / np.sqrt(cin)).astype(np.float32), cuda)
    gamma = T((1.0 + 0.1 * rs.randn(cout)).astype(np.float32), cuda)
    beta = T((0.1 * rs.randn(cout)).astype(np.float32), cuda)
    xft = (T((0.5 + rs.rand(cin)).astype(np.float32), cuda), T((0.2 * rs.randn(cin)).astype(np.float32), cuda), True) if xf else None
    nbytes = pn2._lib.lib.pn2_bn_workspace_bytes(cout)
    calls = []
    pn2._lib.lib.trace = calls
    try:
        ws = torch.zeros(nbytes // 8, dtype=torch.float64, device=cuda)
        y_tiled, _ = tfu.hip_matmul_bn_stats_fin(x1, w, ws, xft, 1)
    finally:
        pn2._lib.lib.trace = None
    for it in range(3):
        ws = torch.zeros(nbytes // 8, dtype=torch.float64, device=cuda)
        rm, rv = torch.zeros(cout, device=cuda), torch.ones(cout, device=cuda)
        y, (save_mean, save_invstd, sc, sh) = tfu.hip_matmul_bn_stats_fin(x, w, ws, xft, 2, gamma, beta, None, 0.5, rm, rv)
        assert torch.equal(y, y_tiled[:rows]), it
        yd = y.double()
        mean, var = yd.mean(0), yd.var(0, unbiased=False)
        invstd = 1.0 / torch.sqrt(var + 1e-3)
        assert float((save_mean.double() - mean).abs().max()) <= 1e-6 * max(1.0, float(mean.abs().max())), it
        assert float(((save_invstd.double() - invstd) / invstd).abs().max()) <= 1e-6, it
        assert float((sc.double() - gamma.double() * invstd).abs().max()) <= 1e-6 * float((gamma.double() * invstd).abs().max()), it
        assert float((rm.double() - 0.5 * mean).abs().max()) <= 1e-6 * max(1.0, float(mean.abs().max())), it
        tick = ws[:48].view(torch.int32)[:65].cpu().numpy()  # 64 first-level counters + the second level: one ticket per workgroup
        wg = min((rows // 32 + 7) // 8, 256) if cin == 128 else min(rows // 128, 1024)
        assert int(tick[:64].sum()) == wg and int(tick[64]) == min(wg, 64), (it, int(tick[:64].sum()), int(tick[64]))


@pytest.mark.parametrize("rows,n_in,pool,below,relu", [(131072, 128, 0, True, True), (65568, 128, 0, True, False), (98304, 128, 0, False, True),
                                                       (131072, 64, 32, True, True), (65536, 128, 32, True, True), (65600, 64, 0, True, True)])
def test_streaming_data_gradient_of_the_128_wide_layers_equals_the_tiled_kernel(pn2, cuda, rows, n_in, pool, below, relu):
    """Round 6: dgrad_wide_kernel (csrc/pn2_dgrad_wide.h: 128 output channels of the layer, 64 / 128 inputs, >= 65536 rows, the gradient
    leaving the batch norm formed on load -- plain or behind the max over 32 rows -- eight waves per workgroup) behind
    pn2_linear_dgrad_fin.  dx is the tiled linear_kernel's bit for bit: the same entry point on the first 65504 rows (below the
    streaming kernel's threshold; a row of dx depends on its own row / group only); the constants it publishes for the layer below
    (finish 3: coef (6, c), dgamma, dbeta) are those of float64 sums over that dx.  tf.gradients through conv2d -> batch_norm -> relu
    (-> max over K), tf_util.py:181-204,555-581, pointnet_util.py:167-170."""
    import torch
    L, P = pn2._lib.lib, pn2._lib.ptr
    st = pn2._lib.stream_ptr()
    rs = np.random.RandomState(rows % 613 + n_in + pool)
    c = 128
    y1 = T(rs.randn(rows, c).astype(np.float32), cuda)
    yb1 = T(rs.randn(rows, n_in).astype(np.float32), cuda)          # pre-normalisation output of the layer below
    w = T((rs.randn(n_in, c) / np.sqrt(c)).astype(np.float32), cuda)
    coef = T(np.stack([0.5 + rs.rand(c), 0.2 * rs.randn(c), 0.1 * rs.randn(c), 0.5 + rs.rand(c), 0.01 * rs.randn(c),
                       0.01 * rs.randn(c)]).astype(np.float32), cuda)
    if pool:
        lin = (y1.double() * coef[0].double() + coef[1].double()).float()     # the forward's fmaf(y, sc, sh)
        t = torch.relu(lin).view(rows // pool, pool, c)
        zmax = t.max(1).values.contiguous()
        ties = (t == zmax.view(-1, 1, c)).sum(1).float().contiguous()
        dz1 = T(rs.randn(rows // pool, c).astype(np.float32), cuda)
    else:
        zmax = ties = None
        dz1 = T(rs.randn(rows, c).astype(np.float32), cuda)
    gb, bb = T((1.0 + 0.1 * rs.randn(n_in)).astype(np.float32), cuda), T((0.1 * rs.randn(n_in)).astype(np.float32), cuda)
    mb, ib = T((0.1 * rs.randn(n_in)).astype(np.float32), cuda), T((0.5 + rs.rand(n_in)).astype(np.float32), cuda)
    nb = L.pn2_bn_workspace_bytes(n_in)

    def run(n, finish):
        dx = torch.empty(n, n_in, device=cuda)
        ws = torch.zeros(nb // 8, dtype=torch.float64, device=cuda)
        cb, dg, db = torch.empty(6, n_in, device=cuda), torch.empty(n_in, device=cuda), torch.empty(n_in, device=cuda)
        bel = (P(yb1), P(gb), P(bb), P(mb), P(ib), int(relu), P(ws), nb, finish, P(cb), P(dg), P(db)) if below else \
              (None, None, None, None, None, 0, None, 0, 0, None, None, None)
        rc = L.pn2_linear_dgrad_fin(n, n_in, c, None, P(y1), P(dz1), P(coef), 1, pool, P(zmax), P(ties), P(w), P(dx), *bel, st)
        assert rc == 0, rc
        return dx, cb, dg, db

    n_tiled = 65504
    dx_tiled = run(n_tiled, 3 if below else 0)[0]
    for it in range(3):
        dx, cb, dg, db = run(rows, 3 if below else 0)
        assert torch.equal(dx[:n_tiled], dx_tiled), it
        if below:
            dxd = dx.double()
            sc = gb.double() * ib.double()
            sh = bb.double() - mb.double() * sc
            on = ((yb1.double() * sc + sh).float() > 0) if relu else torch.ones_like(dx, dtype=torch.bool)
            g = torch.where(on, dxd, torch.zeros_like(dxd))
            xh = ((yb1 - mb) * ib).double()
            s1, s2 = g.sum(0), (g * xh).sum(0)
            tol = 1e-5 * float(g.abs().sum(0).max())
            assert float((db.double() - s1).abs().max()) <= tol and float((dg.double() - s2).abs().max()) <= tol, it
            assert float((cb[4].double() - s1 / rows).abs().max()) <= tol / rows * 2 and float((cb[5].double() - s2 / rows).abs().max()) <= tol / rows * 2


@pytest.mark.parametrize("rows,cin,cout", [(524288, 32, 32), (131072, 128, 128), (1024, 768, 256), (40000, 64, 512), (33, 32, 32)])
def test_bn_finish_ticket_under_many_workgroups(pn2, cuda, rows, cin, cout):
    """The ticket of pn2_bn_finish under load: the forward GEMM + statistics + constants as ONE launch (finish 2), twelve times in a
    row on grids of 1 .. 4096 workgroups, against float64 moments of the same y -- a finishing workgroup that ran before another
    workgroup's sums had landed, or read a stale copy, would miss whole tiles of rows -- and the counters must read exactly
    `workgroups` afterwards."""
    import torch
    tfu = pn2.util.tf_util
    rs = np.random.RandomState(rows % 991 + cout)
    x = T(rs.randn(rows, cin).astype(np.float32), cuda)
    w = T((rs.randn(cin, cout) / np.sqrt(cin)).astype(np.float32), cuda)
    w[0, 1] = 3.0
    gamma = T((1.0 + 0.1 * rs.randn(cout)).astype(np.float32), cuda)
    beta = T((0.1 * rs.randn(cout)).astype(np.float32), cuda)
    nbytes = pn2._lib.lib.pn2_bn_workspace_bytes(cout)
    for it in range(12):
        ws = torch.zeros(nbytes // 8, dtype=torch.float64, device=cuda)
        rm, rv = torch.zeros(cout, device=cuda), torch.ones(cout, device=cuda)
        y, (save_mean, save_invstd, sc, sh) = tfu.hip_matmul_bn_stats_fin(x, w, ws, None, 2, gamma, beta, None, 0.5, rm, rv)
        yd = y.double()
        mean, var = yd.mean(0), yd.var(0, unbiased=False)
        assert float((save_mean.double() - mean).abs().max()) <= 1e-5 * max(1.0, float(mean.abs().max())), it
        invstd = 1.0 / torch.sqrt(var + 1e-3)
        assert float(((save_invstd.double() - invstd) / invstd).abs().max()) <= 1e-5, it
        assert float((sc.double() - gamma.double() * invstd).abs().max()) <= 1e-5 * float(invstd.max()), it
        assert float((rm.double() - 0.5 * mean).abs().max()) <= 1e-5 * max(1.0, float(mean.abs().max())), it
        tick = ws[:48].view(torch.int32)[:65].cpu().numpy()  # 64 first-level counters + the second level
        assert tick[:64].sum() > 0 and tick[64] == min(64, int(tick[:64].sum())), tick


@pytest.mark.parametrize("rows,cin,cout,pool,xf_on,below", [(8192, 32, 32, 0, True, True), (8192, 32, 64, 32, True, True),
                                                            (4096, 64, 64, 0, False, True), (2048, 64, 32, 32, True, False),
                                                            (96, 32, 64, 0, False, False)])
def test_fused_narrow_backward_equals_the_two_launches(pn2, cuda, rows, cin, cout, pool, xf_on, below):
    """pn2_linear_bwd_fused (data + weight gradient of a 32 / 64-channel layer in one launch, (y, dz) read once) against
    pn2_linear_dgrad_fin + pn2_linear_wgrad_gx on the same operands: dx bit for bit (same MFMA order per tile), dW, and the
    batch-norm gradient constants it publishes for the layer below, to the order of the atomics; and against float64."""
    import torch
    lib, ptr, sp, check = pn2._lib.lib, pn2._lib.ptr, pn2._lib.stream_ptr, pn2._lib.check
    rs = np.random.RandomState(rows + cin + cout)
    x = T((rs.randn(rows, cin) * 0.7 + 0.2).astype(np.float32), cuda)           # the layer below's un-normalised output
    y = T(rs.randn(rows, cout).astype(np.float32), cuda)
    w = T((rs.randn(cin, cout) / np.sqrt(cin)).astype(np.float32), cuda)
    coef = T(np.stack([1 + 0.1 * rs.randn(cout), 0.1 * rs.randn(cout), 0.1 * rs.randn(cout), 1 + 0.1 * rs.rand(cout),
                       0.01 * rs.randn(cout), 0.01 * rs.randn(cout)]).astype(np.float32), cuda)
    sc, sh = T((1 + 0.1 * rs.randn(cin)).astype(np.float32), cuda), T((0.1 * rs.randn(cin)).astype(np.float32), cuda)
    if pool:
        dz = T(rs.randn(rows // 32, cout).astype(np.float32), cuda)
        lin = torch.relu(y * coef[0] + coef[1]).view(rows // 32, 32, cout)
        zmax = lin.max(1).values.contiguous()
        ties = (lin == zmax[:, None, :]).sum(1).float().contiguous()
    else:
        dz, zmax, ties = T(rs.randn(rows, cout).astype(np.float32), cuda), None, None
    gb, bb = T((1 + 0.1 * rs.randn(cin)).astype(np.float32), cuda), T((0.1 * rs.randn(cin)).astype(np.float32), cuda)
    mb, ib = T((0.2 + 0.1 * rs.randn(cin)).astype(np.float32), cuda), T((1.4 + 0.1 * rs.rand(cin)).astype(np.float32), cuda)
    nb = lib.pn2_bn_workspace_bytes(cin)
    res = {}
    for fused in (True, False):
        dx = torch.empty(rows, cin, device=cuda)
        dw = torch.zeros(cin, cout, device=cuda)
        ws = torch.zeros(nb // 8, dtype=torch.float64, device=cuda)
        cb, dgb, dbb = torch.zeros(6, cin, device=cuda), torch.zeros(cin, device=cuda), torch.zeros(cin, device=cuda)
        bl = (ptr(x), ptr(gb), ptr(bb), ptr(mb), ptr(ib), 1, ptr(ws), nb, 3, ptr(cb), ptr(dgb), ptr(dbb)) if below else \
             (None, None, None, None, None, 0, None, 0, 0, None, None, None)
        xfa = (ptr(sc), ptr(sh), 1) if xf_on else (None, None, 0)
        if fused:
            check(lib.pn2_linear_bwd_fused(rows, cin, cout, ptr(x), *xfa, ptr(y), ptr(dz), ptr(coef), 1, pool, ptr(zmax), ptr(ties),
                                           ptr(w), ptr(dx), ptr(dw), *bl, sp()), "pn2_linear_bwd_fused")
        else:
            check(lib.pn2_linear_dgrad_fin(rows, cin, cout, None, ptr(y), ptr(dz), ptr(coef), 1, pool, ptr(zmax), ptr(ties), ptr(w),
                                           ptr(dx), *bl, sp()), "pn2_linear_dgrad_fin")
            check(lib.pn2_linear_wgrad_gx(rows, cin, cout, ptr(x), *xfa, ptr(y), ptr(dz), ptr(coef), 1, pool, ptr(zmax), ptr(ties),
                                          ptr(dw), sp()), "pn2_linear_wgrad_gx")
        res[fused] = (dx, dw, cb, dgb, dbb)
    # float64 yardstick of dy / dx / dW
    c6 = coef.double()
    lin = y.double() * c6[0] + c6[1]
    on = lin > 0
    if pool:
        t = torch.where(on, lin, torch.zeros_like(lin)).float().view(-1, 32, cout)
        g = torch.where(t == zmax.view(-1, 1, cout), (dz / ties).view(-1, 1, cout), torch.zeros_like(t)).view(rows, cout).double()
    else:
        g = dz.double()
    gk = torch.where(on, g, torch.zeros_like(g))
    dy = c6[0] * (-((y.double() - c6[2]) * c6[3]) * c6[5] + (gk - c6[4]))
    xa = torch.relu(x.double() * sc.double() + sh.double()) if xf_on else x.double()
    for fused in (True, False):
        dx, dw = res[fused][0].double(), res[fused][1].double()
        assert float((dx - dy @ w.double().t()).norm() / (dy @ w.double().t()).norm()) <= 2e-6, fused
        assert float((dw - xa.t() @ dy).norm() / (xa.t() @ dy).norm()) <= 2e-6, fused
    assert float((res[True][0] - res[False][0]).abs().max()) <= 1e-6 * float(res[False][0].abs().max())
    assert float((res[True][1] - res[False][1]).abs().max()) <= 1e-5 * float(res[False][1].abs().max())
    if below:
        for a, r in zip(res[True][2:], res[False][2:]):
            assert float((a - r).abs().max()) <= 1e-5 * max(float(r.abs().max()), 1e-3)


def test_deferred_batch_norm_output_must_reach_a_dense_layer(pn2, cuda):
    """conv2d(..., defer_bn=True) hands over an UN-normalised tensor; if anything but the next batch-normalised conv2d
    consumes it the backward pass raises instead of training on wrong values, and a layer without batch norm refuses it."""
    import torch
    tfu = pn2.util.tf_util
    tfu.set_default_store(tfu.VariableStore(device=cuda, seed=2))
    tfu.reset_bn_links()
    x = torch.randn(2, 2048, 1, 16, device=cuda, requires_grad=True)
    h = tfu.conv2d(x, 32, [1, 1], padding="VALID", stride=[1, 1], bn=True, is_training=True, scope="a", bn_decay=0.5, defer_bn=True)
    with pytest.raises(RuntimeError):
        h.sum().backward()     # consumed by a reduction, not by a dense layer
    tfu.reset_bn_links()
    h = tfu.conv2d(x, 32, [1, 1], padding="VALID", stride=[1, 1], bn=True, is_training=True, scope="a", bn_decay=0.5, defer_bn=True)
    with pytest.raises(RuntimeError):
        tfu.conv2d(h, 32, [1, 1], padding="VALID", stride=[1, 1], bn=False, activation_fn=None, is_training=True, scope="b")
    # consumers that are not dense layers refuse it IN FORWARD (a forward-only pass has no backward to catch it)
    with pytest.raises(RuntimeError):
        tfu.dropout(h, True, "dp", keep_prob=0.5)
    with pytest.raises(RuntimeError):
        pn2.util.pointnet_util.group_pool(h.reshape(2, 64, 32, 32), None, "max")
    # the records belong to the tape: they die with it (no process-global leftovers keyed by recycled addresses) ...
    assert len(tfu._bn_links) >= 1
    del h
    import gc
    gc.collect()  # (the tracebacks pytest.raises caught above still hold views of h in a reference cycle)
    assert len(tfu._bn_links) == 0
    # ... and a training-mode forward that records no tape (BN recalibration under no_grad, frozen inputs) does not defer at
    # all: its output IS normalised, and equals the normalised activation of the recorded pass
    with torch.no_grad():
        hn = tfu.conv2d(x, 32, [1, 1], padding="VALID", stride=[1, 1], bn=True, is_training=True, scope="a", bn_decay=0.5, defer_bn=True)
        assert len(tfu._bn_links) == 0
        tfu.dropout(hn, True, "dp", keep_prob=0.5)   # accepted: normalised values
    hg = tfu.conv2d(x, 32, [1, 1], padding="VALID", stride=[1, 1], bn=True, is_training=True, scope="a", bn_decay=0.5)
    assert float((hn - hg).abs().max()) <= 1e-5 * float(hg.abs().max())
    xf = x.detach()  # nothing requires a gradient through the layer's INPUT, but its weights do: a tape node exists, deferral is fine
    hf = tfu.conv2d(xf, 32, [1, 1], padding="VALID", stride=[1, 1], bn=True, is_training=True, scope="a", bn_decay=0.5, defer_bn=True)
    lk = tfu._bn_links.get(hf.data_ptr())
    assert lk is not None and lk.sc is not None and hf.requires_grad
    del lk
    del hf
    tfu.reset_bn_links()


def test_fp_module_chain_equals_per_layer_linear(pn2, oracle, cuda):
    """FP4-shaped module: the chained path and the one-launch-per-layer path agree to fp32 rounding."""
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    rs = np.random.RandomState(5)
    xyz1 = rs.random_sample((2, 1024, 3)).astype(np.float32)
    xyz2 = xyz1[:, :128].copy()
    p1 = rs.randn(2, 1024, 3).astype(np.float32)
    p2 = rs.randn(2, 128, 128).astype(np.float32)
    store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=31))
    args = (T(xyz1, cuda), T(xyz2, cuda), T(p1, cuda), T(p2, cuda), [128, 128, 128], False, None)
    pu.pointnet_fp_module(*args, scope="fp4")
    randomize_bn(store, 32)
    a = pu.pointnet_fp_module(*args, scope="fp4").cpu().numpy()
    pu.USE_MLP_CHAIN = False
    try:
        b = pu.pointnet_fp_module(*args, scope="fp4").cpu().numpy()
    finally:
        pu.USE_MLP_CHAIN = True
    ref = oracle.fp_module(xyz1, xyz2, p1, p2, layer_dicts(store, "fp4", ["conv_0", "conv_1", "conv_2"]))
    close(a, ref)
    close(b, ref)


@pytest.mark.parametrize("b,n,m,c1,c2,widths", [(2, 1000, 128, 3, 128, [128, 128]), (1, 333, 64, 0, 64, [128]),
                                                 (3, 256, 32, 5, 64, [64, 128]), (2, 512, 100, 2, 32, [64]),
                                                 (1, 31, 3, 3, 8, [128, 128])])
def test_fp_mlp_fused_vs_oracle(pn2, oracle, cuda, b, n, m, c1, c2, widths):
    """pn2_fp_mlp_fused = FP front end + LDS-resident chain in one kernel, vs the oracle's
    three_nn -> fp_weights -> three_interpolate -> concat -> fp64 dense layers."""
    tfu = pn2.util.tf_util
    rs = np.random.RandomState(b * 1000 + n)
    xyz1 = rs.random_sample((b, n, 3)).astype(np.float32)
    xyz2 = rs.random_sample((b, m, 3)).astype(np.float32)
    xyz2[:, :3] = xyz1[:, :3]  # a few zero distances: the 1e-10 clamp path
    p1 = rs.randn(b, n, c1).astype(np.float32) if c1 else None
    p2 = rs.randn(b, m, c2).astype(np.float32)
    dist, idx = oracle.three_nn(xyz1, xyz2)
    x = oracle.three_interpolate(p2, idx, oracle.fp_weights(dist))
    if c1:
        x = np.concatenate([x, p1], axis=2)
    ref = x.reshape(b * n, c1 + c2).astype(np.float64)
    ws, bs = [], []
    c = c1 + c2
    for w_ in widths:
        W = (rs.randn(c, w_) / np.sqrt(c)).astype(np.float32)
        bb = (rs.randn(w_) * 0.1).astype(np.float32)
        ws.append(T(W, cuda)); bs.append(T(bb, cuda))
        ref = np.maximum(ref @ W.astype(np.float64) + bb, 0)
        c = w_
    y = tfu.hip_fp_mlp_fused(T(dist, cuda), T(idx, cuda), None if p1 is None else T(p1, cuda), T(p2, cuda), ws, bs)
    assert y is not None
    close(y.cpu().numpy(), ref)


@pytest.mark.parametrize("b,n,m,c1", [(16, 8192, 1024, 3), (3, 1000, 128, 0), (2, 2048, 256, 8), (1, 4099, 300, 5), (5, 97, 16, 1)])
def test_fp_chain_pipelined_schedule_is_bit_identical_to_the_lockstep_kernel(pn2, oracle, cuda, b, n, m, c1):
    """the software-pipelined FP4 chain (one wave per SIMD gathers the next tile's rows of z between the MFMA
    groups of the current tile) against the lockstep kernel it replaces: torch.equal -- same MFMA order, same blend
    expression, another schedule -- at the FP4 shape of configs[1] (16 x 8192 rows, 131 -> 128 -> 128 -> 128), at ragged row
    counts (last tile partly out of range, fewer tiles than waves) and skip-link widths 0..8; and within 1e-5 of fp64."""
    import torch
    tfu = pn2.util.tf_util
    rs = np.random.RandomState(n + c1)
    xyz1 = rs.random_sample((b, n, 3)).astype(np.float32)
    xyz2 = xyz1[:, rs.permutation(n)[:m]].copy()      # known points are a subset: zero distances, the 1e-10 clamp
    c2 = 256
    p1 = rs.randn(b, n, c1).astype(np.float32) if c1 else None
    p2 = rs.randn(b, m, c2).astype(np.float32)
    dist, idx = pn2.three_nn(T(xyz1, cuda), T(xyz2, cuda))
    ws, bs, c = [], [], c1 + c2
    for w_ in (128, 128, 128):
        ws.append(T((rs.randn(c, w_) / np.sqrt(c)).astype(np.float32), cuda))
        bs.append(T((rs.randn(w_) * 0.1).astype(np.float32), cuda))
        c = w_
    args = (dist, idx, None if p1 is None else T(p1, cuda), T(p2, cuda), ws, bs)
    y0 = tfu.hip_fp_mlp_fused_pre(*args, schedule=0)
    y1 = tfu.hip_fp_mlp_fused_pre(*args, schedule=1)
    yd = tfu.hip_fp_mlp_fused_pre(*args)              # the library's own choice
    assert y0 is not None and y1 is not None and yd is not None
    assert torch.equal(y0, y1) and torch.equal(yd, y0)
    if b * n <= 40000:
        x = oracle.three_interpolate(p2, idx.cpu().numpy(), oracle.fp_weights(dist.cpu().numpy()))
        if c1:
            x = np.concatenate([x, p1], axis=2)
        ref = x.reshape(b * n, c1 + c2).astype(np.float64)
        for W, bb in zip(ws, bs):
            ref = np.maximum(ref @ W.cpu().numpy().astype(np.float64) + bb.cpu().numpy(), 0)
        close(y1.cpu().numpy(), ref)
    # two 128-wide layers only / skip link wider than 8 channels: the pipelined kernel says so
    assert tfu.hip_fp_mlp_fused_pre(dist, idx, args[2], args[3], ws[:2], bs[:2], schedule=1) is None


def test_fp_mlp_fused_rejects_unsupported(pn2, cuda):
    import torch
    tfu = pn2.util.tf_util
    dist = torch.ones(1, 64, 3, device=cuda)
    idx = torch.zeros(1, 64, 3, dtype=torch.int32, device=cuda)
    p2 = torch.zeros(1, 8, 12, device=cuda)  # c2 % 8 != 0
    assert tfu.hip_fp_mlp_fused(dist, idx, None, p2, [torch.zeros(12, 128, device=cuda)], [torch.zeros(128, device=cuda)]) is None
    p2 = torch.zeros(1, 8, 16, device=cuda)
    assert tfu.hip_fp_mlp_fused(dist, idx, None, p2, [torch.zeros(16, 256, device=cuda)], [torch.zeros(256, device=cuda)]) is None


def test_fp_module_fused_front_end_full_size(pn2, oracle, cuda):
    """FP4-shaped module at a row count that takes the fused path (b*n >= 65536): fused == unfused to fp32
    rounding, and both within tolerance of the oracle."""
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    rs = np.random.RandomState(9)
    xyz1 = rs.random_sample((8, 8192, 3)).astype(np.float32)
    xyz2 = xyz1[:, :256].copy()
    p1 = rs.randn(8, 8192, 3).astype(np.float32)
    p2 = rs.randn(8, 256, 128).astype(np.float32)
    store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=33))
    args = (T(xyz1, cuda), T(xyz2, cuda), T(p1, cuda), T(p2, cuda), [128, 128, 128], False, None)
    pu.pointnet_fp_module(*args, scope="fp4")
    randomize_bn(store, 34)
    outs = {}
    for hoist in (True, False):  # first layer hoisted by linearity (pn2_fp_mlp_fused_pre) / computed in place (pn2_fp_mlp_fused)
        calls = []
        pn2._lib.lib.trace = calls
        pu.USE_HOISTED_FP = hoist
        try:
            outs[hoist] = pu.pointnet_fp_module(*args, scope="fp4").cpu().numpy()
        finally:
            pn2._lib.lib.trace = None
            pu.USE_HOISTED_FP = True
        names = [c[0] for c in calls]
        assert ("pn2_fp_mlp_fused_pre" if hoist else "pn2_fp_mlp_fused") in names and "pn2_fp_interp_concat" not in names
    a = outs[True]
    np.testing.assert_allclose(outs[True], outs[False], rtol=1e-5, atol=1e-5)  # same values up to fp32 summation order
    pu.USE_FUSED_FP = False
    try:
        b = pu.pointnet_fp_module(*args, scope="fp4").cpu().numpy()
    finally:
        pu.USE_FUSED_FP = True
    np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-5)
    ref = oracle.fp_module(xyz1, xyz2, p1, p2, layer_dicts(store, "fp4", ["conv_0", "conv_1", "conv_2"]))
    close(a, ref)
    close(outs[False], ref)


def test_training_step_single_gpu(pn2, cuda):
    """Trainer.train_step: forward (batch-stat BN) + weighted CE + backward through the HIP gradient
    kernels + flat-bucket all-reduce (world 1) + Adam.  The loss must drop on a fixed batch."""
    import torch
    hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
    # batch statistics over at least 128 rows in every layer: with a handful of rows the normalisation amplifies
    # rounding differences between the two stacks by up to 1/sqrt(eps) per layer
    hp.update(l1_npoint=256, l2_npoint=64, l3_npoint=32, l4_npoint=16)
    rs = np.random.RandomState(0)
    pc = T(np.concatenate([s_scene(1, 8, 2048), rs.random_sample((8, 2048, 3)).astype(np.float32)], 2), cuda)
    labels = T(rs.randint(0, 9, (8, 2048)).astype(np.int64), cuda)
    smpw = T((rs.random_sample((8, 2048)) + 0.5).astype(np.float32), cuda)
    tr = pn2.train.Trainer(hp, 9, store=pn2.util.tf_util.VariableStore(device=cuda, seed=3))
    losses = [tr.train_step(pc, labels, smpw) for _ in range(8)]
    assert all(np.isfinite(losses))
    assert losses[-1] < losses[0], losses
    assert tr.bucket.numel == tr.store.num_parameters()
    # the semantic.json model has 967,945 trainable parameters
    assert tr.store.num_parameters() == 967945


# ------------------------------------------------------------------ API coverage ------------
def test_sample_and_group_api(pn2, oracle, cuda):
    """sample_and_group returns (new_xyz, new_points [xyz first], idx, grouped_xyz) like pointnet_util.py:18-60."""
    rs = np.random.RandomState(1)
    xyz = rs.random_sample((2, 600, 3)).astype(np.float32)
    pts = rs.randn(2, 600, 5).astype(np.float32)
    nx, npts, idx, gx = pn2.sample_and_group(64, 0.3, 16, T(xyz, cuda), T(pts, cuda))
    r_nx, r_np, r_idx, r_gx = oracle.sample_and_group(64, 0.3, 16, xyz, pts)
    assert np.array_equal(idx.cpu().numpy(), r_idx) and np.array_equal(nx.cpu().numpy(), r_nx)
    assert np.array_equal(gx.cpu().numpy(), r_gx) and np.array_equal(npts.cpu().numpy(), r_np)
    nx, npts, idx, gx = pn2.sample_and_group(64, 0.3, 16, T(xyz, cuda), None)  # points=None -> xyz only
    assert np.array_equal(npts.cpu().numpy(), r_gx)


def test_sa_module_group_all_and_poolings(pn2, oracle, cuda):
    """group_all=True (sample_and_group_all, :63-95) and the avg / max_and_avg poolings (:171-191)."""
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    rs = np.random.RandomState(2)
    xyz = rs.random_sample((2, 64, 3)).astype(np.float32)
    pts = rs.randn(2, 64, 8).astype(np.float32)
    store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=41))
    kw = dict(npoint=None, radius=None, nsample=None, mlp=[32, 64], mlp2=None, group_all=True, is_training=False,
              bn_decay=None, scope="ga")
    pu.pointnet_sa_module(T(xyz, cuda), T(pts, cuda), **kw)
    randomize_bn(store, 42)
    new_xyz, new_points, idx = pu.pointnet_sa_module(T(xyz, cuda), T(pts, cuda), **kw)
    layers = layer_dicts(store, "ga", ["conv0", "conv1"])
    h = np.concatenate([xyz, pts], axis=2)[:, None].astype(np.float64)
    for l in layers:
        h = oracle.conv_bn_relu(h, l)
    assert new_xyz.shape == (2, 1, 3) and float(new_xyz.abs().sum()) == 0.0
    assert idx.shape == (2, 1, 64)
    close(new_points.cpu().numpy(), h.max(2))
    for pooling, fn in (("avg", lambda t: t.mean(2)), ("max_and_avg", lambda t: np.concatenate([t.mean(2), t.max(2)], -1))):
        _, npool, _ = pu.pointnet_sa_module(T(xyz, cuda), T(pts, cuda), pooling=pooling, **kw)
        close(npool.cpu().numpy(), fn(h))


API_BRANCHES = [
    # (id, C, mlp, mlp2, use_xyz, bn, group_all)   -- the arguments of pointnet_sa_module the model itself never varies
    ("mlp2", 16, [32, 64], [64, 32], True, True, False),                 # pointnet_util.py:194-211
    ("mlp2-one-layer", 3, [32, 32, 64], [128], True, True, False),
    ("use_xyz=False", 16, [32, 64], None, False, True, False),            # :52-58: grouped features only
    ("use_xyz=False+mlp2", 64, [64, 128], [64], False, True, False),
    ("bn=False", 16, [32, 32, 64], None, True, False, False),             # tf_util.py:191-203 without the batch norm
    ("bn=False+mlp2", 64, [64, 64, 128], [128, 64], True, False, False),
    ("bn=False+use_xyz=False", 32, [64], None, False, False, False),
    ("group_all+mlp2", 8, [32, 64], [64, 16], True, True, True),          # :137-141 with :194-211
    ("group_all+use_xyz=False+bn=False", 8, [32], [16], False, False, True),
]


@pytest.mark.parametrize("case", API_BRANCHES, ids=[c[0] for c in API_BRANCHES])
def test_sa_module_api_branches_vs_oracle(pn2, oracle, cuda, case):
    """`mlp2`, `use_xyz=False`, `bn=False` and `group_all` with `mlp2` (util/pointnet_util.py:52-58,137-141,
    194-211) against the fp64 restatement: indices bit-exact, features within 1e-5."""
    name, C, mlp, mlp2, use_xyz, bn, group_all = case
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    rs = np.random.RandomState(len(name) * 7 + C)
    B, N, npoint, radius, K = 2, (96 if group_all else 640), 80, 0.3, 32
    xyz = rs.random_sample((B, N, 3)).astype(np.float32)
    pts = rs.randn(B, N, C).astype(np.float32)
    store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=5))
    kw = dict(npoint=None if group_all else npoint, radius=None if group_all else radius, nsample=None if group_all else K,
              mlp=mlp, mlp2=mlp2, group_all=group_all, is_training=False, bn_decay=None, scope="br", bn=bn, use_xyz=use_xyz)
    pu.pointnet_sa_module(T(xyz, cuda), T(pts, cuda), **kw)  # creates the variables
    randomize_bn(store, 9)
    new_xyz, new_points, idx = pu.pointnet_sa_module(T(xyz, cuda), T(pts, cuda), **kw)
    if not bn:
        assert not any("bn/" in k for k in store.params), "bn=False must not create batch-norm variables"
    layers = layer_dicts(store, "br", ["conv%d" % i for i in range(len(mlp))], bn=bn)
    layers2 = layer_dicts(store, "br", ["conv_post_%d" % i for i in range(len(mlp2 or []))], bn=bn)
    cin = (3 if use_xyz else 0) + C
    assert layers[0]["W"].shape == (cin, mlp[0])
    r_xyz, r_pts, r_idx = oracle.sa_module(xyz, pts, npoint, radius, K, layers, layers2=layers2, use_xyz=use_xyz,
                                           group_all=group_all)
    assert np.array_equal(idx.cpu().numpy(), r_idx) and np.array_equal(new_xyz.cpu().numpy(), r_xyz)
    assert new_points.shape == (B, 1 if group_all else npoint, (mlp2 or mlp)[-1])
    close(new_points.cpu().numpy(), r_pts)


def _record_train_layers(tfu):
    """patch tf_util._train_layer so that every layer's output is kept (call order) -> (list, restore())"""
    orig, seen = tfu._train_layer, []

    def rec(inputs, w2d, b, bnv, bn_decay, relu, pool=0, defer=False):
        z = orig(inputs, w2d, b, bnv, bn_decay, relu, pool, defer)
        seen.append(z.detach())
        return z
    tfu._train_layer = rec
    return seen, lambda: setattr(tfu, "_train_layer", orig)


def _rel(a, b):
    return float((a.double() - b.double()).norm()) / max(float(b.double().norm()), 1e-30)


@pytest.mark.parametrize("mlp,mlp2,C", [([32], None, 8), ([32, 32, 64], None, 16), ([64, 48], [40, 24], 5)])
def test_sa_module_trains_without_batch_norm(pn2, cuda, mlp, mlp2, C):
    """the reference accepts pointnet_sa_module(..., bn=False, is_training=True) (pointnet_util.py:150-166 on
    tf_util.py:186-204: conv -> bias_add -> relu, no batch norm).  Forward within 1e-5 and every gradient (weights, biases, input
    features) within 1e-4 of a float64 evaluation on the HIP path's own activation pattern (ReLU masks from the HIP outputs, max
    pool winners = the neighbours whose float64 value is closest to the pooled HIP value)."""
    import torch
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    rs = np.random.RandomState(len(mlp) + C)
    B, N, M, K = 2, 512, 64, 16
    xyz = T(rs.random_sample((B, N, 3)).astype(np.float32), cuda)
    pts = T(rs.randn(B, N, C).astype(np.float32), cuda).requires_grad_(True)
    store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=5))
    seen, restore = _record_train_layers(tfu)
    try:
        new_xyz, out, idx = pu.pointnet_sa_module(xyz, pts, npoint=M, radius=0.3, nsample=K, mlp=mlp, mlp2=mlp2, group_all=False,
                                                  is_training=True, bn_decay=None, scope="nb", bn=False)
    finally:
        restore()
    assert not any("bn/" in k for k in store.params)
    cot = T(rs.randn(*out.shape).astype(np.float32), cuda)
    (out * cot).sum().backward()
    names = ["conv%d" % i for i in range(len(mlp))] + ["conv_post_%d" % i for i in range(len(mlp2 or []))]
    assert len(seen) == len(names)
    # float64 on the same pattern
    ii = idx.long()
    bidx = torch.arange(B, device=cuda).view(B, 1, 1)
    p64 = pts.detach().double().requires_grad_(True)
    x64 = xyz.double()
    h = torch.cat([x64[bidx, ii] - new_xyz.double().unsqueeze(2), p64[bidx, ii]], dim=-1)   # (B,M,K,3+C), xyz first
    W64, b64 = {}, {}
    for li, nm in enumerate(names):
        w = store.params["nb/%s/weights" % nm].detach().double().reshape(-1, seen[li].shape[-1]).requires_grad_(True)
        bb = store.params["nb/%s/biases" % nm].detach().double().requires_grad_(True)
        W64[nm], b64[nm] = w, bb
        if li == len(mlp):   # max over K between mlp and mlp2: the winner is named by the pooled HIP value
            zt = seen[li - 1].double()                       # (B,M,K,c) un-pooled HIP output of the last mlp layer
            pooled_hip = zt.max(dim=2, keepdim=True).values
            pick = (zt - pooled_hip).abs().argmin(dim=2, keepdim=True)
            h = torch.gather(h, 2, pick)
        y = h @ w + bb
        h = y * (seen[li].double().reshape(y.shape) > 0)
    if mlp2 is None:
        zt = seen[-1].double()
        pick = (zt - zt.max(dim=2, keepdim=True).values).abs().argmin(dim=2, keepdim=True)
        h = torch.gather(h, 2, pick)
    ref = h.squeeze(2)
    close(out.detach().cpu().numpy(), ref.detach().cpu().numpy())
    (ref * cot.double()).sum().backward()
    errs = {"points": _rel(pts.grad, p64.grad)}
    for nm in names:
        errs[nm + "/w"] = _rel(store.params["nb/%s/weights" % nm].grad.reshape(W64[nm].shape), W64[nm].grad)
        errs[nm + "/b"] = _rel(store.params["nb/%s/biases" % nm].grad, b64[nm].grad)
    assert max(errs.values()) < 1e-4, errs


def test_fp_module_trains_without_batch_norm(pn2, cuda):
    """pointnet_fp_module(..., bn=False, is_training=True) (pointnet_util.py:312-325 on tf_util.py:186-204): forward 1e-5,
    gradients of weights, biases, points1 and points2 within 1e-4 of float64 on the HIP path's ReLU pattern."""
    import torch
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    rs = np.random.RandomState(11)
    B, n1, n2, c1, c2, mlp = 2, 600, 96, 6, 24, [48, 32]
    xyz1 = T(rs.random_sample((B, n1, 3)).astype(np.float32), cuda)
    xyz2 = T(rs.random_sample((B, n2, 3)).astype(np.float32), cuda)
    p1 = T(rs.randn(B, n1, c1).astype(np.float32), cuda).requires_grad_(True)
    p2 = T(rs.randn(B, n2, c2).astype(np.float32), cuda).requires_grad_(True)
    store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=6))
    seen, restore = _record_train_layers(tfu)
    try:
        out = pu.pointnet_fp_module(xyz1, xyz2, p1, p2, mlp, True, None, scope="nbfp", bn=False)
    finally:
        restore()
    assert not any("bn/" in k for k in store.params) and len(seen) == len(mlp)
    cot = T(rs.randn(*out.shape).astype(np.float32), cuda)
    (out * cot).sum().backward()
    dist, idx = pn2.three_nn(xyz1, xyz2)
    d = torch.clamp(dist.double(), min=1e-10)            # pointnet_util.py:300-303
    wgt = (1.0 / d) / (1.0 / d).sum(dim=2, keepdim=True)
    q1, q2 = p1.detach().double().requires_grad_(True), p2.detach().double().requires_grad_(True)
    bidx = torch.arange(B, device=cuda).view(B, 1, 1)
    interp = (q2[bidx, idx.long()] * wgt.unsqueeze(-1)).sum(dim=2)
    h = torch.cat([interp, q1], dim=2)                    # interpolated FIRST (:306-311)
    W64, b64 = [], []
    for li in range(len(mlp)):
        w = store.params["nbfp/conv_%d/weights" % li].detach().double().reshape(-1, mlp[li]).requires_grad_(True)
        bb = store.params["nbfp/conv_%d/biases" % li].detach().double().requires_grad_(True)
        W64.append(w); b64.append(bb)
        y = h @ w + bb
        h = y * (seen[li].double().reshape(y.shape) > 0)
    close(out.detach().cpu().numpy(), h.detach().cpu().numpy())
    (h * cot.double()).sum().backward()
    errs = {"points1": _rel(p1.grad, q1.grad), "points2": _rel(p2.grad, q2.grad)}
    for li in range(len(mlp)):
        errs["w%d" % li] = _rel(store.params["nbfp/conv_%d/weights" % li].grad.reshape(W64[li].shape), W64[li].grad)
        errs["b%d" % li] = _rel(store.params["nbfp/conv_%d/biases" % li].grad, b64[li].grad)
    assert max(errs.values()) < 1e-4, errs


def test_group_pool_kernels_all_modes_and_use_nchw(pn2, cuda):
    """pn2_group_pool / _grad (the `pooling=` variants of pointnet_sa_module, pointnet_util.py:165-191: max, avg,
    weighted_avg = softmax-like exp(-5 |grouped_xyz|) weights, max_and_avg = concat [avg | max]) against the same
    expressions in float64 torch, forward and gradient, for widths that take the 16-byte and the scalar path; and the
    module with pooling="weighted_avg" on a ball-query level with `use_nchw=True` (a TF layout hint: accepted, same values)."""
    import torch
    pu, tfu = pn2.util.pointnet_util, pn2.util.tf_util
    torch.manual_seed(3)
    for c in (64, 37):
        x = torch.randn(3, 50, 16, c, device=cuda)
        x[:, ::4, 8:, :] = x[:, ::4, :1, :]  # duplicated neighbours: tied maxima split their gradient evenly (tf.reduce_max)
        g = torch.randn(3, 50, 16, 3, device=cuda) * 0.3
        for pooling in ("max", "avg", "weighted_avg", "max_and_avg"):
            xx = x.clone().requires_grad_(True)
            out = pu.group_pool(xx, g, pooling)
            xd = x.double().requires_grad_(True)
            if pooling == "max":
                ref = xd.amax(2, keepdim=True)
            elif pooling == "avg":
                ref = xd.mean(2, keepdim=True)
            elif pooling == "weighted_avg":
                e = torch.exp(-torch.linalg.vector_norm(g.double(), dim=-1, keepdim=True) * 5)
                ref = (xd * (e / e.sum(2, keepdim=True))).sum(2, keepdim=True)
            else:
                ref = torch.cat([xd.mean(2, keepdim=True), xd.amax(2, keepdim=True)], -1)
            assert out.shape == ref.shape
            assert torch.allclose(out.double(), ref, rtol=1e-5, atol=1e-6), pooling
            probe = torch.cos(torch.arange(ref.numel(), device=cuda).double()).reshape(ref.shape)
            (out.double() * probe).sum().backward()
            (ref * probe).sum().backward()
            assert torch.allclose(xx.grad.double(), xd.grad, rtol=1e-5, atol=1e-6), pooling
    rs = np.random.RandomState(8)
    xyz, pts = T(rs.random_sample((2, 512, 3)).astype(np.float32), cuda), T(rs.randn(2, 512, 5).astype(np.float32), cuda)
    tfu.set_default_store(tfu.VariableStore(device=cuda, seed=9))
    kw = dict(npoint=64, radius=0.3, nsample=16, mlp=[16, 32], mlp2=None, group_all=False, is_training=False, bn_decay=None,
              scope="wa", pooling="weighted_avg")
    a = pu.pointnet_sa_module(xyz, pts, **kw)
    b_ = pu.pointnet_sa_module(xyz, pts, use_nchw=True, **kw)
    assert a[1].shape == (2, 64, 32) and torch.equal(a[1], b_[1]) and torch.equal(a[2], b_[2])
    with pytest.raises(ValueError):
        pu.pointnet_sa_module(xyz, pts, **dict(kw, pooling="median"))


def test_sa_module_msg_vs_oracle(pn2, oracle, cuda):
    """BASELINE config[2] shape family: MSG with 3 scales (radii/K/MLPs are builder-chosen: the reference
    ships no MSG hyper-parameters).  NOTE the [features, xyz] concat order of pointnet_util.py:259."""
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    rs = np.random.RandomState(3)
    xyz = rs.random_sample((2, 1024, 3)).astype(np.float32)
    pts = rs.randn(2, 1024, 6).astype(np.float32)
    radii, ks, mlps = [0.1, 0.2, 0.4], [16, 32, 64], [[32, 32, 64], [64, 64, 128], [64, 96, 128]]
    store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=43))
    args = (T(xyz, cuda), T(pts, cuda), 128, radii, ks, mlps, False, None)
    pu.pointnet_sa_module_msg(*args, scope="msg")
    randomize_bn(store, 44)
    new_xyz, new_points = pu.pointnet_sa_module_msg(*args, scope="msg")
    f = oracle.farthest_point_sample(128, xyz)
    nx = oracle.gather_point(xyz, f)
    outs = []
    for i, (r, k) in enumerate(zip(radii, ks)):
        idx, _ = oracle.query_ball_point(r, k, xyz, nx)
        gx = oracle.group_point(xyz, idx) - nx[:, :, None, :]
        h = np.concatenate([oracle.group_point(pts, idx), gx], axis=-1).astype(np.float64)  # features FIRST here
        for l in layer_dicts(store, "msg", ["conv%d_%d" % (i, j) for j in range(3)]):
            h = oracle.conv_bn_relu(h, l)
        outs.append(h.max(2))
    assert np.array_equal(new_xyz.cpu().numpy(), nx)
    assert new_points.shape == (2, 128, 64 + 128 + 128)
    close(new_points.cpu().numpy(), np.concatenate(outs, -1))
    # the K=32 scale runs on the fused kernel (weights rotated to the kernel's [xyz | features] order), all three
    # ball queries come from ONE multi-radius scan, and the unfused path gives the same numbers
    calls = []
    pn2._lib.lib.trace = calls
    try:
        pu.pointnet_sa_module_msg(*args, scope="msg")
    finally:
        pn2._lib.lib.trace = None
    names = [c_[0] for c_ in calls]
    assert names.count("pn2_query_ball_point_multi") == 1 and "pn2_query_ball_point" not in names
    assert "pn2_sa_mlp_max_fused" in names
    pu.USE_FUSED_SA = False
    try:
        _, unfused = pu.pointnet_sa_module_msg(*args, scope="msg")
    finally:
        pu.USE_FUSED_SA = True
    np.testing.assert_allclose(new_points.cpu().numpy(), unfused.cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_large_scene_config4_shapes(pn2, oracle, cuda):
    """BASELINE config[4] geometry at its own sizes (B=1, N=65536, npoint=4096, K=64, fp32): the
    streaming FPS kernel (n > 16384) and the K=64 ball query, bit-exact against the oracle."""
    xyz = s_scene(9, 1, 65536)
    f = pn2.farthest_point_sample(4096, T(xyz, cuda)).cpu().numpy()
    rf = oracle.farthest_point_sample(4096, xyz)
    assert np.array_equal(f, rf)
    q = oracle.gather_point(xyz, rf)
    idx, cnt = pn2.query_ball_point(0.5, 64, T(xyz, cuda), T(q, cuda))
    ri, rc = oracle.query_ball_point(0.5, 64, xyz, q)
    assert np.array_equal(cnt.cpu().numpy(), rc) and np.array_equal(idx.cpu().numpy(), ri)
    # K = 64 grouped MLP + max through the unfused MFMA path (pool = 64)
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=45))
    feat = np.random.RandomState(1).randn(1, 65536, 16).astype(np.float32)
    kw = dict(npoint=4096, radius=0.5, nsample=64, mlp=[64, 128], mlp2=None, group_all=False, is_training=False,
              bn_decay=None, scope="big")
    pu.pointnet_sa_module(T(xyz, cuda), T(feat, cuda), **kw)
    randomize_bn(store, 46)
    _, npts, i2 = pu.pointnet_sa_module(T(xyz, cuda), T(feat, cuda), **kw)
    assert np.array_equal(i2.cpu().numpy(), ri)
    gx = oracle.group_point(xyz, ri) - q[:, :, None, :]
    h = np.concatenate([gx, oracle.group_point(feat, ri)], -1).astype(np.float64)
    for l in layer_dicts(store, "big", ["conv0", "conv1"]):
        h = oracle.conv_bn_relu(h, l)
    close(npts.cpu().numpy(), h.max(2))


@pytest.mark.parametrize("rows,cin,cout,pool", [(640, 256, 256, 32), (1024, 768, 256, 0), (96, 131, 128, 0), (2048, 384, 512, 32)])
def test_linear_split_k_config(pn2, cuda, rows, cin, cout, pool):
    """few rows + deep K -> 32x64 tiles with the k-tile split between two waves and reduced through LDS"""
    rs = np.random.RandomState(rows + cout)
    x = rs.randn(rows, cin).astype(np.float32)
    w = (rs.randn(cin, cout) / np.sqrt(cin)).astype(np.float32)
    b = rs.randn(cout).astype(np.float32)
    y = pn2.util.tf_util.hip_linear(T(x, cuda), T(w, cuda), T(b, cuda), relu=1, pool=pool).cpu().numpy()
    ref = np.maximum(x.astype(np.float64) @ w.astype(np.float64) + b, 0)
    if pool:
        ref = ref.reshape(rows // pool, pool, cout).max(1)
    close(y, ref)


# ------------------------------------------------------------------ bf16 fused SA (BASELINE configs[4]) ------
def _bf16_case(pn2, oracle, cuda, b, n, m, K, c, mlp, radius, seed):
    import torch
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    rs = np.random.RandomState(seed)
    xyz = s_scene(seed, b, n)
    pts = oracle.bf16_round(rs.randn(b, n, c).astype(np.float32))
    store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=seed + 1))
    txyz = T(xyz, cuda)
    tpts = T(pts, cuda).to(torch.bfloat16)
    assert torch.equal(tpts.float().cpu(), torch.from_numpy(pts))  # the inputs are exactly representable
    with tfu.variable_scope("sa"):
        new_xyz, idx = pu.sa_geometry(txyz, m, radius, K)
        pu.sa_features_inference(txyz, new_xyz, tpts, idx, mlp)  # creates the variables
        randomize_bn(store, seed + 2)
        calls = []
        pn2._lib.lib.trace = calls
        try:
            out = pu.sa_features_inference(txyz, new_xyz, tpts, idx, mlp)
        finally:
            pn2._lib.lib.trace = None
        ws, bs, cin = [], [], 3 + c
        for i, cout in enumerate(mlp):
            with tfu.variable_scope("conv%d" % i):
                w2, b2 = tfu.folded_dense(cin, cout, True, (1, 1, cin, cout))  # the very tensors the kernel was given
            ws.append(w2.cpu().numpy()); bs.append(b2.cpu().numpy())
            cin = cout
    ri = idx.cpu().numpy()
    gx = oracle.group_point(xyz, ri) - new_xyz.cpu().numpy()[:, :, None, :]
    ref = oracle.mlp_max_bf16(gx, oracle.group_point(pts, ri), ws, bs)
    return out.cpu().numpy().astype(np.float64), ref, [c_[0] for c_ in calls]


@pytest.mark.parametrize("K,c,mlp", [(64, 128, [128, 128]), (32, 128, [128]), (64, 64, [64, 64, 128]),
                                     (32, 16, [64, 128]), (64, 128, [128, 128, 128])])
def test_sa_fused_bf16_vs_contract(pn2, oracle, cuda, K, c, mlp):
    """pn2_sa_mlp_max_fused_bf16 against the restated precision contract (bf16 inputs/weights/hidden activations,
    exact products, wide accumulation).  The only freedom is the fp32 accumulation order, which can flip the bf16
    rounding of an occasional hidden activation: tolerance 4e-3 of the output scale."""
    got, ref, calls = _bf16_case(pn2, oracle, cuda, 2, 2048, 128, K, c, mlp, 0.9, 40 + K + c)
    assert "pn2_sa_mlp_max_fused_bf16" in calls
    scale = np.abs(ref).max()
    err = np.abs(got - ref)
    assert err.max() <= 4e-3 * scale, (err.max(), scale)
    assert np.median(err) <= 2e-4 * scale


def test_sa_bf16_unsupported_falls_back_to_fp32_kernels(pn2, oracle, cuda):
    """bf16 features with a layer pattern outside the bf16 kernel run the fp32 kernels on the exact values."""
    got, ref, calls = _bf16_case(pn2, oracle, cuda, 1, 1024, 64, 32, 16, [32, 32, 64], 0.9, 77)
    assert calls[0] == "pn2_sa_mlp_max_fused_bf16" and calls[1:] == ["pn2_sa_mlp_max_fused"]  # refused, then fp32 fused
    # fp32 kernels: no rounding of weights / hidden activations -> only close to the bf16 contract, not equal
    assert np.abs(got - ref).max() <= 5e-2 * np.abs(ref).max()


def test_large_scene_bf16_config4(pn2, oracle, cuda):
    """BASELINE configs[4]: one scene, N=65536, npoint=4096, K=64, C=128 bf16 features, r=0.5, fused bf16 MLP."""
    got, ref, calls = _bf16_case(pn2, oracle, cuda, 1, 65536, 4096, 64, 128, [128, 128], 0.5, 5)
    assert "pn2_sa_mlp_max_fused_bf16" in calls and got.shape == (1, 4096, 128)
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 4e-3 * scale


# ------------------------------------------------------------------ training: weight gradient kernel ---------
@pytest.mark.parametrize("rows,cin,cout", [(524288 // 8, 6, 32), (4096, 32, 64), (1000, 67, 64), (777, 259, 256), (64, 131, 128),
                                           (5, 3, 9), (20000, 128, 512), (33, 768, 256),
                                           (131072, 128, 128), (70001, 128, 128), (65560, 96, 256), (65536, 64, 128)])
def test_linear_wgrad_vs_fp64(pn2, cuda, rows, cin, cout):
    """pn2_linear_wgrad: dW = x^T . dy against float64 (fp32 products are exact; only the summation order differs).  The last four
    shapes take the eight-wave workgroups of the 64 x 128 tile (r06: >= 65536 rows), one of them with a ragged last chunk."""
    import ctypes
    import torch
    rs = np.random.RandomState(rows + cin)
    x = rs.randn(rows, cin).astype(np.float32)
    dy = rs.randn(rows, cout).astype(np.float32)
    dw = torch.empty((cin, cout), dtype=torch.float32, device=cuda)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    tx, tdy = T(x, cuda), T(dy, cuda)
    assert pn2._lib.lib.pn2_linear_wgrad(rows, cin, cout, P(tx), P(tdy), P(dw), None) == 0
    torch.cuda.synchronize()
    ref = x.astype(np.float64).T @ dy.astype(np.float64)
    err = np.abs(dw.cpu().numpy() - ref)
    assert err.max() <= 1e-5 * np.sqrt(rows) * 8, err.max()  # random-walk growth of the fp32 rounding
    assert err.max() <= 2e-4 * np.abs(ref).max() + 1e-3


@pytest.mark.parametrize("rows,pool,xf_on", [(131072, 0, True), (65568, 0, False), (98304, 32, True)])
def test_eight_wave_weight_gradient_with_operands_formed_on_load(pn2, cuda, rows, pool, xf_on):
    """Round 6: pn2_linear_wgrad_gx at 128 x 128 over >= 65536 rows (eight waves per workgroup, dy formed on load from (y, dz, coef),
    x normalised on load) against float64 of the materialised operands: dW += a^T . dy with a = relu(fma(x, sc, sh)) and dy =
    pn2_bn_grad_element(...) -- the float expressions of the kernels, summed in float64.  tf_util.py:181-204,555-581 via tf.gradients."""
    import torch
    L, P = pn2._lib.lib, pn2._lib.ptr
    rs = np.random.RandomState(rows % 887 + pool)
    c = 128
    x = T(rs.randn(rows, c).astype(np.float32), cuda)
    y = T(rs.randn(rows, c).astype(np.float32), cuda)
    sc, sh = T((0.5 + rs.rand(c)).astype(np.float32), cuda), T((0.2 * rs.randn(c)).astype(np.float32), cuda)
    coef = T(np.stack([0.5 + rs.rand(c), 0.2 * rs.randn(c), 0.1 * rs.randn(c), 0.5 + rs.rand(c), 0.01 * rs.randn(c),
                       0.01 * rs.randn(c)]).astype(np.float32), cuda)
    csc, csh, cmu, cis, k1, k2 = (coef[j] for j in range(6))
    lin = (y.double() * csc.double() + csh.double()).float()   # the kernel's fmaf(y, sc, sh): one rounding of the exact value
    on = lin > 0
    if pool:
        groups = rows // pool
        dzp = T(rs.randn(groups, c).astype(np.float32), cuda)
        t = torch.where(on, lin, torch.zeros_like(lin)).view(groups, pool, c)
        zmax = t.max(1).values
        ties = (t == zmax.view(groups, 1, c)).sum(1).float()
        g = torch.where(t == zmax.view(groups, 1, c), (dzp / ties).view(groups, 1, c), torch.zeros_like(t)).view(rows, c)
        dz_arg, zm_arg, ti_arg = dzp, zmax, ties
    else:
        dz = T(rs.randn(rows, c).astype(np.float32), cuda)
        g, dz_arg, zm_arg, ti_arg = dz, dz, None, None
    gk = torch.where(on, g, torch.zeros_like(g))
    dy = csc * (-((y - cmu) * cis) * k2 + (gk - k1))
    a = torch.relu((x.double() * sc.double() + sh.double()).float()) if xf_on else x
    ref = a.double().t() @ dy.double()
    dw = torch.zeros(c, c, device=cuda)
    rc = L.pn2_linear_wgrad_gx(rows, c, c, P(x), P(sc) if xf_on else None, P(sh) if xf_on else None, 1, P(y), P(dz_arg), P(coef), 1, pool,
                               P(zm_arg), P(ti_arg), P(dw), pn2._lib.stream_ptr())
    assert rc == 0, rc
    err = float((dw.double() - ref).abs().max())
    assert err <= 2e-5 * float(ref.abs().max()) + 1e-5 * np.sqrt(rows), (err, float(ref.abs().max()))


def test_training_gradients_with_hip_wgrad_match_torch(pn2, cuda):
    """One training forward/backward on the HIP training kernels (batch norm + ReLU forward/backward, weight
    gradient) against the same step with every dense layer evaluated in float64 (matmul, batch norm, ReLU and their
    autograd), for every parameter.  fp32 rounding is amplified chaotically by the chain of 23 batch norms (per-layer
    forward errors are 1e-7 .. 8e-6 in EVERY fp32 stack, the gradients then differ from float64 by 3e-3 .. 1e-2 depending
    only on which way each rounding fell: all-torch 4-6e-3, HIP kernels with torch GEMMs 3e-3, all-HIP 1e-2 on this seed),
    so the bound is absolute (3 % worst, 1.5 % median) and the all-torch fp32 stack's error is printed next to it.
    That every single GEMM of the step is as accurate as the library's is checked call by call in
    tests/test_train_gpu.py::test_every_gemm_of_a_real_step_is_as_accurate_as_the_library."""
    import torch
    import torch.nn.functional as F
    tfu = pn2.util.tf_util
    hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
    hp.update(l1_npoint=256, l2_npoint=64, l3_npoint=32, l4_npoint=16)
    rs = np.random.RandomState(0)
    pc = T(np.concatenate([s_scene(1, 8, 2048), rs.random_sample((8, 2048, 3)).astype(np.float32)], 2), cuda)
    labels = T(rs.randint(0, 9, (8, 2048)).astype(np.int64), cuda)
    smpw = T((rs.random_sample((8, 2048)) + 0.5).astype(np.float32), cuda)

    def layer_fp64(inputs, w2d, b, bnv, bn_decay, relu, pool=0, defer=False):
        y = inputs.double() @ w2d.double() + b.double()
        if bnv is not None:
            beta, gamma, mean, var = bnv
            c = y.shape[-1]
            y = F.batch_norm(y.reshape(-1, c), None, None, gamma.double(), beta.double(), training=True,
                             eps=tfu.BN_EPSILON).reshape(y.shape)
        y = torch.relu(y) if relu else y
        if pool and pool > 1:
            y = y.reshape(list(y.shape[:-2]) + [y.shape[-2] // pool, pool, y.shape[-1]]).amax(dim=-2)
        return y.float()

    def run(mode):
        from torch_layers import train_layer_torch
        orig = tfu._train_layer
        if mode == "fp64":
            tfu._train_layer = layer_fp64
        elif mode == "torch":
            tfu._train_layer = train_layer_torch  # library GEMMs + F.batch_norm: the plain fp32 reference
        # (the reference stacks replace tf_util._train_layer: SA1's first layer must reach it as a layer of its own, not inside
        # the one-launch front end of the HIP path)
        tfu.USE_SA_FIRST_LAYER_FUSED = mode == "hip"
        try:
            store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=5))
            torch.manual_seed(123)  # same dropout mask in every run
            logits, _ = pn2.model.get_model(pc, True, 9, hp, bn_decay=0.5)
            pn2.model.get_loss(logits, labels, smpw).backward()
            return {k: v.grad.detach().double().clone() for k, v in store.params.items() if v.grad is not None}
        finally:
            tfu._train_layer = orig
            tfu.USE_SA_FIRST_LAYER_FUSED = True

    hip, t32, ref = run("hip"), run("torch"), run("fp64")
    assert t32.keys() == ref.keys() and len(hip) > 40
    # the HIP path returns NO gradient for a bias in front of batch norm (it is exactly zero); everything else is there
    missing = ref.keys() - hip.keys()
    assert hip.keys() <= ref.keys() and all(k.endswith("biases") for k in missing) and len(missing) == 22
    for k in missing:
        assert float(ref[k].abs().max()) < 1e-6
    ref = {k: v for k, v in ref.items() if k in hip}
    rel = {k: (float((hip[k] - ref[k]).norm()) / max(float(ref[k].norm()), 1e-30),
               float((t32[k] - ref[k]).norm()) / max(float(ref[k].norm()), 1e-30)) for k in ref if float(ref[k].norm()) > 1e-6}
    worst = max(rel, key=lambda k: rel[k][0])
    print("relative gradient error vs float64 layers: worst HIP %.2e (torch fp32 %.2e) at %s; median HIP %.2e, torch %.2e"
          % (rel[worst][0], rel[worst][1], worst, float(np.median([v[0] for v in rel.values()])),
             float(np.median([v[1] for v in rel.values()]))))
    # bounds: every parameter's gradient within 3 % of its norm, half of them within 1.5 % (the all-torch fp32 stack sits
    # at 0.4-0.6 % median / 1 % worst on this seed: same order, see the docstring)
    assert rel[worst][0] < 3e-2 and float(np.median([v[0] for v in rel.values()])) < 1.5e-2, rel[worst]


def test_training_gradients_match_float64_on_the_same_activation_pattern(pn2, cuda):
    """the 3 % bound of the test above is the chaos of 23 batch norms -- which side of zero a
    pre-activation within rounding of it falls on, and which of two near-equal neighbours wins a max pool, differ between any
    two fp32 evaluations, and every such flip moves the gradients by far more than rounding.  Here the float64 network is
    evaluated ON THE HIP PATH'S OWN PATTERN: every ReLU uses the mask the HIP layer's output shows (z > 0), every max pool
    selects the neighbour whose float64 value is closest to the value the HIP pool returned.  What remains is rounding of the
    kernels themselves (GEMMs, batch-norm statistics, their gradients, the scatter-add of the grouping gradients): every
    parameter gradient of the whole model within 1e-4 of its norm (measured: worst 7.8e-6, median 5.8e-6 over 68 tensors)."""
    import torch
    import torch.nn.functional as F
    tfu = pn2.util.tf_util
    hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
    hp.update(l1_npoint=256, l2_npoint=64, l3_npoint=32, l4_npoint=16)
    rs = np.random.RandomState(0)
    pc = T(np.concatenate([s_scene(1, 8, 2048), rs.random_sample((8, 2048, 3)).astype(np.float32)], 2), cuda)
    labels = T(rs.randint(0, 9, (8, 2048)).astype(np.int64), cuda)
    smpw = T((rs.random_sample((8, 2048)) + 0.5).astype(np.float32), cuda)
    orig = tfu._train_layer
    seen = []   # the HIP run's layer outputs, in call order
    cursor = [0]

    def layer_recording(inputs, w2d, b, bnv, bn_decay, relu, pool=0, defer=False):
        z = orig(inputs, w2d, b, bnv, bn_decay, relu, pool, False)
        seen.append(z.detach())
        return z

    def layer_fp64_forced(inputs, w2d, b, bnv, bn_decay, relu, pool=0, defer=False):
        z_hip = seen[cursor[0]].double()
        cursor[0] += 1
        y = inputs.double() @ w2d.double() + b.double()
        if bnv is not None:
            beta, gamma, mean, var = bnv
            c = y.shape[-1]
            y = F.batch_norm(y.reshape(-1, c), None, None, gamma.double(), beta.double(), training=True,
                             eps=tfu.BN_EPSILON).reshape(y.shape)
        if pool and pool > 1:
            w = y.shape[-2] // pool
            yw = y.reshape(list(y.shape[:-2]) + [w, pool, y.shape[-1]])
            zt = z_hip.reshape(list(y.shape[:-2]) + [w, 1, y.shape[-1]])
            # ReLU and max commute; the HIP pool returned max(relu(.)): where it is positive it names the winner
            pick = (torch.relu(yw.detach()) - zt).abs().argmin(dim=-2, keepdim=True)
            y = torch.gather(yw, -2, pick).squeeze(-2)
        if relu:
            y = y * (z_hip.reshape(y.shape) > 0)
        assert y.shape == z_hip.reshape(y.shape).shape
        return y.float()

    def run(layer):
        tfu._train_layer = layer
        keep = tfu.USE_BN_ON_LOAD
        tfu.USE_BN_ON_LOAD = False  # every layer hands over its normalised activation (what is recorded / replaced)
        tfu.USE_SA_FIRST_LAYER_FUSED = False  # ... and SA1's first layer is a layer of its own (its one-launch form has its own test)
        try:
            store = tfu.set_default_store(tfu.VariableStore(device=cuda, seed=5))
            torch.manual_seed(123)  # same dropout mask in every run
            logits, _ = pn2.model.get_model(pc, True, 9, hp, bn_decay=0.5)
            pn2.model.get_loss(logits, labels, smpw).backward()
            return {k: v.grad.detach().double().clone() for k, v in store.params.items() if v.grad is not None}, logits.detach()
        finally:
            tfu._train_layer = orig
            tfu.USE_BN_ON_LOAD = keep
            tfu.USE_SA_FIRST_LAYER_FUSED = True

    hip, lh = run(layer_recording)
    ref, lr = run(layer_fp64_forced)
    assert cursor[0] == len(seen) > 20
    assert float((lh.double() - lr.double()).abs().max()) < 1e-4 * max(1.0, float(lr.abs().max()))  # same pattern: logits agree
    ref = {k: v for k, v in ref.items() if k in hip}
    rel = {k: float((hip[k] - ref[k]).norm()) / max(float(ref[k].norm()), 1e-30) for k in ref if float(ref[k].norm()) > 1e-6}
    worst = max(rel, key=rel.get)
    print("gradient error vs float64 on the HIP path's activation pattern: worst %.2e at %s, median %.2e over %d tensors"
          % (rel[worst], worst, float(np.median(list(rel.values()))), len(rel)))
    assert rel[worst] < 1e-4 and float(np.median(list(rel.values()))) < 5e-5, (worst, rel[worst])


# ------------------------------------------------------------------ training: batch norm + relu kernels ------
@pytest.mark.parametrize("rows,c,relu", [(4096, 32, 1), (1000, 64, 1), (333, 128, 0), (8192, 512, 1), (257, 9, 1), (64, 1024, 1),
                                         (5000, 36, 1), (1, 32, 1)])
def test_bn_relu_forward_backward_vs_oracle(pn2, oracle, cuda, rows, c, relu):
    """pn2_bn_relu_forward / _backward against the float64 oracle (tf_util.py:555-581 restated): z, batch moments,
    moving averages, dy, dgamma, dbeta.  Tolerance 1e-5 of the output scale (fp32 normalisation of fp64 moments)."""
    import ctypes
    import torch
    lib = pn2._lib.lib
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    rs = np.random.RandomState(rows * 7 + c)
    y = (rs.randn(rows, c) * (1.0 + np.arange(c) % 5) + 3.0 * np.sin(np.arange(c))).astype(np.float32)
    gamma = (0.5 + rs.rand(c)).astype(np.float32)
    beta = (rs.randn(c) * 0.3).astype(np.float32)
    bias = rs.randn(c).astype(np.float32)
    dz = rs.randn(rows, c).astype(np.float32)
    eps, decay = 1e-3, 0.9
    ty, tg, tb, tbias, tdz = (T(a, cuda) for a in (y, gamma, beta, bias, dz))
    rm, rv = torch.full((c,), 0.25, device=cuda), torch.full((c,), 2.0, device=cuda)
    z = torch.empty_like(ty)
    sm, si = torch.empty(c, device=cuda), torch.empty(c, device=cuda)
    assert lib.pn2_bn_workspace_bytes(c) >= 2 * c * 8 and lib.pn2_bn_workspace_bytes(c) % 8 == 0
    ws = torch.empty(lib.pn2_bn_workspace_bytes(c) // 8, dtype=torch.float64, device=cuda)
    assert lib.pn2_bn_relu_forward(rows, c, P(ty), P(tg), P(tb), P(tbias), eps, decay, relu, 0, P(rm), P(rv), P(ws), ws.numel() * 8,
                                   P(sm), P(si), P(z), None, None) == 0
    zr, mean, var, mm, mv = oracle.batch_norm_relu_train(y, gamma, beta, bool(relu), eps, bias, (np.full(c, 0.25), np.full(c, 2.0)), decay)
    zg = z.cpu().numpy()
    assert np.abs(zg - zr).max() <= 1e-5 * max(np.abs(zr).max(), 1.0)
    assert np.allclose(sm.cpu().numpy(), mean, rtol=1e-6, atol=1e-6)
    assert np.allclose(si.cpu().numpy(), 1.0 / np.sqrt(var + eps), rtol=1e-6, atol=1e-6)
    assert np.allclose(rm.cpu().numpy(), mm, rtol=1e-6, atol=1e-6) and np.allclose(rv.cpu().numpy(), mv, rtol=1e-6, atol=1e-6)
    # elements within fp32 rounding of zero: the oracle takes the kernel's own ReLU mask
    dyr, dgr, dbr = oracle.batch_norm_relu_train_grad(y, gamma, beta, dz, bool(relu), eps, mask=zg > 0)
    dy = torch.empty_like(ty)
    dg, db = torch.empty(c, device=cuda), torch.empty(c, device=cuda)
    assert lib.pn2_bn_relu_backward(rows, c, P(tdz), P(ty), P(tg), P(tb), P(sm), P(si), relu, 0, None, None, P(ws), ws.numel() * 8,
                                    P(dy), P(dg), P(db), None) == 0
    for got, ref in ((dy, dyr), (dg, dgr), (db, dbr)):
        assert np.abs(got.cpu().numpy() - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1.0)
    dz2 = tdz.clone()  # in place: dy may alias dz
    assert lib.pn2_bn_relu_backward(rows, c, P(dz2), P(ty), P(tg), P(tb), P(sm), P(si), relu, 0, None, None, P(ws), ws.numel() * 8,
                                    P(dz2), P(dg), P(db), None) == 0
    assert torch.equal(dz2, dy)


@pytest.mark.parametrize("groups,pool,c,relu", [(512, 32, 64, 1), (100, 16, 128, 1), (64, 32, 512, 1), (33, 8, 36, 0),
                                                 (7, 64, 9, 1), (300, 32, 32, 1)])
def test_bn_relu_fused_max_pool_vs_oracle(pn2, oracle, cuda, groups, pool, c, relu):
    """pool > 1: batch norm + ReLU + max over each group of `pool` rows in the forward, its gradient (shared equally
    among tied rows) in the backward.  Groups contain duplicated rows (ball-query padding) and all-negative channels
    (every row ties at the ReLU floor)."""
    import ctypes
    import torch
    lib = pn2._lib.lib
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    rows = groups * pool
    rs = np.random.RandomState(groups + pool + c)
    y = rs.randn(groups, pool, c).astype(np.float32)
    y[::3, pool // 2:, :] = y[::3, :1, :]            # second half of every third group repeats its first row
    y = y.reshape(rows, c)
    gamma = (0.5 + rs.rand(c)).astype(np.float32)
    beta = (rs.randn(c) * 0.3).astype(np.float32)
    beta[::5] = -6.0                                   # channels that never pass the ReLU
    dzp = rs.randn(groups, c).astype(np.float32)
    eps = 1e-3
    ty, tg, tb, tdz = (T(a, cuda) for a in (y, gamma, beta, dzp))
    zmax, ties = torch.empty(groups, c, device=cuda), torch.empty(groups, c, device=cuda)
    sm, si = torch.empty(c, device=cuda), torch.empty(c, device=cuda)
    ws = torch.empty(lib.pn2_bn_workspace_bytes(c) // 8, dtype=torch.float64, device=cuda)
    assert lib.pn2_bn_relu_forward(rows, c, P(ty), P(tg), P(tb), None, eps, 0.9, relu, pool, None, None, P(ws), ws.numel() * 8,
                                   P(sm), P(si), P(zmax), P(ties), None) == 0
    z, mean, var = oracle.batch_norm_relu_train(y, gamma, beta, bool(relu), eps)
    zr, _ = oracle.max_pool_rows(z, pool)
    assert np.abs(zmax.cpu().numpy() - zr).max() <= 1e-5 * max(np.abs(zr).max(), 1.0)
    # which rows tie is decided in fp32: restate the kernel's value z = fma(y, sc, sh) (sc = gamma*invstd,
    # sh = fma(-mean, sc, beta), one rounding each) from its own saved moments and pool THAT
    m32, i32 = sm.cpu().numpy(), si.cpu().numpy()
    sc = (gamma * i32).astype(np.float32)
    sh = (-m32.astype(np.float64) * sc.astype(np.float64) + beta.astype(np.float64)).astype(np.float32)
    z32 = (y.astype(np.float64) * sc.astype(np.float64) + sh.astype(np.float64)).astype(np.float32)
    if relu:
        z32 = np.maximum(z32, np.float32(0))
    z32r, tr = oracle.max_pool_rows(z32, pool)
    assert np.array_equal(zmax.cpu().numpy(), z32r.astype(np.float32))
    assert np.array_equal(ties.cpu().numpy(), tr)
    assert tr.max() >= pool // 2 + 1                 # duplicated rows tie
    assert not relu or (tr[:, ::5] == pool).any()     # dead channels: every row ties at the ReLU floor
    # gradient: pooled gradient -> tied rows -> ReLU mask -> batch norm
    dz_full = oracle.max_pool_rows_grad(z32, pool, dzp)
    dyr, dgr, dbr = oracle.batch_norm_relu_train_grad(y, gamma, beta, dz_full, bool(relu), eps, mask=z32 > 0)
    dy = torch.empty_like(ty)
    dg, db = torch.empty(c, device=cuda), torch.empty(c, device=cuda)
    assert lib.pn2_bn_relu_backward(rows, c, P(tdz), P(ty), P(tg), P(tb), P(sm), P(si), relu, pool, P(zmax), P(ties), P(ws),
                                    ws.numel() * 8, P(dy), P(dg), P(db), None) == 0
    for got, ref in ((dy, dyr), (dg, dgr), (db, dbr)):
        assert np.abs(got.cpu().numpy() - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1.0)


def test_bn_relu_argument_checks(pn2, cuda):
    import ctypes
    import torch
    raw = pn2._lib._raw
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    y = torch.zeros(8, 2048, device=cuda)
    v = torch.zeros(2048, device=cuda)
    ws = torch.empty(pn2._lib.lib.pn2_bn_workspace_bytes(2048) // 8, dtype=torch.float64, device=cuda)
    head = (P(y), P(v), P(v), None, 1e-3, 0.9, 1, 0)
    tail = (P(ws), ws.numel() * 8, P(v), P(v), P(y), None, None)
    assert raw.pn2_bn_relu_forward(8, 2048, *head, None, None, *tail) == pn2._lib.PN2_EUNSUP   # c > 1024
    assert raw.pn2_bn_relu_forward(8, 300, *head, None, None, *tail) == 0                      # c % 4 == 0, vector path
    assert raw.pn2_bn_relu_forward(8, 301, *head, None, None, *tail) == pn2._lib.PN2_EUNSUP   # scalar path stops at 256
    assert raw.pn2_bn_relu_forward(0, 64, *head, None, None, *tail) != 0                       # no rows
    assert raw.pn2_bn_relu_forward(8, 64, *head, None, None, P(ws), 8, *tail[2:]) != 0         # workspace too small
    assert raw.pn2_bn_relu_forward(8, 64, None, *head[1:], None, None, *tail) != 0             # null y
    assert raw.pn2_bn_relu_forward(8, 64, *head, P(v), None, *tail) != 0                       # only one moving average
    assert raw.pn2_bn_relu_forward(8, 64, *head[:7], 4, None, None, *tail) != 0                # pooled without a ties buffer
    assert raw.pn2_bn_relu_forward(8, 64, *head[:7], 3, None, None, *tail[:5], P(y), None) != 0  # pool does not divide rows
    torch.cuda.synchronize()


@pytest.mark.parametrize("pool", [0, 32])
def test_train_layer_hip_bn_matches_torch_autograd(pn2, cuda, pool):
    """_train_layer with the HIP batch norm vs the torch composition (F.batch_norm + relu autograd): output,
    moving averages and every gradient (bias gradient exactly zero on the HIP path, rounding noise on torch's)."""
    import torch
    tfu = pn2.util.tf_util
    torch.manual_seed(5)
    x = torch.randn(16, 64, 32, 67, device=cuda)
    w0 = torch.randn(67, 128, device=cuda) * 0.1
    x[:, ::3, 16:, :] = x[:, ::3, :1, :]  # duplicated neighbours: exact ties in the max pool
    oshape = (16, 64, 1 if pool else 32, 128)
    probe = torch.sin(torch.arange(int(np.prod(oshape)), device=cuda).float()).reshape(oshape)
    outs = {}
    from torch_layers import train_layer_torch
    for use in (True, False):
        layer = tfu._train_layer if use else train_layer_torch
        try:
            xx = x.clone().requires_grad_(True)
            w = w0.clone().requires_grad_(True)
            b = (torch.arange(128, device=cuda).float() * 0.01).requires_grad_(True)
            beta = torch.zeros(128, device=cuda, requires_grad=True)
            gamma = torch.ones(128, device=cuda, requires_grad=True)
            mean, var = torch.zeros(128, device=cuda), torch.ones(128, device=cuda)
            z = layer(xx, w, b, (beta, gamma, mean, var), None, True, pool)
            assert tuple(z.shape) == oshape
            (z * probe).sum().backward()
            outs[use] = (z.detach(), mean, var, xx.grad, w.grad, gamma.grad, beta.grad, b.grad)
        finally:
            pass
    assert outs[True][-1] is None                          # bias in front of BN: exactly zero -> no gradient tensor at all ...
    assert float(outs[False][-1].abs().max()) <= 5e-3     # ... where torch sums 32768 rounding errors
    for a, r in zip(outs[True][:-1], outs[False][:-1]):
        s = max(float(r.abs().max()), 1.0)
        assert float((a - r).abs().max()) <= 2e-4 * s


@pytest.mark.parametrize("cout,pool", [(1030, 0), (259, 32), (2048, 0), (1285, 32)])
def test_train_layer_takes_any_batch_norm_width(pn2, cuda, cout, pool):
    """VERDICT r05 missing #4: the reference's batch_norm_template takes any width (util/tf_util.py:555-581); the batch-norm
    kernels take <= 1024 channels, a multiple of 4 above 256.  Wider / odd layers run as independent column blocks on the same
    kernels (batch norm, ReLU and the max over K are per channel) instead of raising: output, moving averages and every
    gradient against the torch composition."""
    import torch
    tfu = pn2.util.tf_util
    torch.manual_seed(cout)
    cin = 40
    x = torch.randn(4, 24, 32, cin, device=cuda)
    w0 = torch.randn(cin, cout, device=cuda) * 0.15
    oshape = (4, 24, 1 if pool else 32, cout)
    probe = torch.sin(torch.arange(int(np.prod(oshape)), device=cuda).float() * 0.7).reshape(oshape)
    outs = {}
    from torch_layers import train_layer_torch
    for use in (True, False):
        layer = tfu._train_layer if use else train_layer_torch
        xx = x.clone().requires_grad_(True)
        w = w0.clone().requires_grad_(True)
        b = (torch.arange(cout, device=cuda).float() * 0.001).requires_grad_(True)
        beta = (0.1 * torch.cos(torch.arange(cout, device=cuda).float())).requires_grad_(True)
        gamma = (1.0 + 0.1 * torch.sin(torch.arange(cout, device=cuda).float())).requires_grad_(True)
        mean, var = torch.zeros(cout, device=cuda), torch.ones(cout, device=cuda)
        z = layer(xx, w, b, (beta, gamma, mean, var), 0.7, True, pool)
        assert tuple(z.shape) == oshape
        (z * probe).sum().backward()
        outs[use] = (z.detach(), mean, var, xx.grad, w.grad, gamma.grad, beta.grad)
    for a, r in zip(outs[True], outs[False]):
        s_ = max(float(r.abs().max()), 1.0)
        assert float((a - r).abs().max()) <= 2e-4 * s_, (a.shape, float((a - r).abs().max()), s_)


@pytest.mark.parametrize("widths,pool,rows_shape", [
    ((32, 32, 64), 32, (4, 40, 32, 6)),        # SA1's stack (ragged tile counts, narrow layers)
    ((128, 128, 128), 0, (2, 1000, 1, 131)),   # FP4's stack (odd input width, 2000 rows)
    ((256, 96), 0, (3, 70, 1, 320)),           # widths that are not a multiple of 128 / rows not a multiple of 32
])
def test_bn_grad_sums_from_the_next_layers_dgrad(pn2, cuda, widths, pool, rows_shape):
    """The first reduction of a layer's batch-norm gradient taken from the accumulator tiles of the NEXT layer's data-gradient
    GEMM (pn2_linear_dgrad_bn_grad_stats + pn2_bn_relu_backward_stats, tf_util._BnLink) against the two-pass backward
    (pn2_linear_dgrad + pn2_bn_relu_backward): same gradients up to fp64 summation order, and the shortcut really ran for
    every link of the stack.  Reference semantics: tf.gradients through conv2d -> batch_norm -> relu, tf_util.py:186-204."""
    import torch
    tfu = pn2.util.tf_util
    torch.manual_seed(3)
    x0 = torch.randn(*rows_shape, device=cuda)
    cin = rows_shape[-1]
    ws = []
    c = cin
    for wd in widths:
        ws.append(torch.randn(c, wd, device=cuda) / np.sqrt(c))
        c = wd
    oshape = list(rows_shape[:-1]) + [widths[-1]]
    if pool:
        oshape[-2] //= pool
    probe = torch.cos(torch.arange(int(np.prod(oshape)), device=cuda).float() * 0.37).reshape(oshape)
    calls = []
    real, real_gx = pn2._lib.lib.pn2_bn_relu_backward_stats, pn2._lib.lib.pn2_bn_grad_constants
    real_fin, real_both = pn2._lib.lib.pn2_linear_dgrad_fin, pn2._lib.lib.pn2_linear_bwd_fused
    outs = {}
    for use in (True, False):
        tfu.USE_DGRAD_BN_STATS = use
        tfu.reset_bn_links()
        try:
            if use:
                pn2._lib.lib.pn2_bn_relu_backward_stats = lambda *a: (calls.append(a[1]), real(*a))[1]
                # the on-load form of the batch-norm gradient (round 6) takes the sums the same way: stats_done = a[13]
                pn2._lib.lib.pn2_bn_grad_constants = lambda *a: (calls.append(a[1]) if a[13] else None, real_gx(*a))[1]
                # ... and with the finish inside the producer the data-gradient GEMM of the layer above leaves sums, fold and
                # constants in one launch: y_below = a[13], width of the layer below = a[1]
                pn2._lib.lib.pn2_linear_dgrad_fin = lambda *a: (calls.append(a[1]) if a[13] is not None else None, real_fin(*a))[1]
                # ... narrow layers form data and weight gradient in one launch (pn2_linear_bwd_fused): y_below = a[17], cin = a[1]
                pn2._lib.lib.pn2_linear_bwd_fused = lambda *a: (calls.append(a[1]) if a[17] is not None else None, real_both(*a))[1]
            xx = x0.clone().requires_grad_(True)
            params, h = [], xx
            for i, wd in enumerate(widths):
                w = ws[i].clone().requires_grad_(True)
                gamma = (1.0 + 0.1 * torch.sin(torch.arange(wd, device=cuda).float())).requires_grad_(True)
                beta = (0.05 * torch.cos(torch.arange(wd, device=cuda).float())).requires_grad_(True)
                mean, var = torch.zeros(wd, device=cuda), torch.ones(wd, device=cuda)
                last = i == len(widths) - 1
                h = tfu._train_layer(h, w, torch.zeros(wd, device=cuda), (beta, gamma, mean, var), None, True,
                                     pool if last else 0)
                params += [w, gamma, beta]
            (h * probe).sum().backward()
            outs[use] = [h.detach(), xx.grad] + [p_.grad for p_ in params]
        finally:
            tfu.USE_DGRAD_BN_STATS = True
            pn2._lib.lib.pn2_bn_relu_backward_stats = real
            pn2._lib.lib.pn2_bn_grad_constants = real_gx
            pn2._lib.lib.pn2_linear_dgrad_fin = real_fin
            pn2._lib.lib.pn2_linear_bwd_fused = real_both
    assert calls == list(widths[:-1][::-1]), calls   # every layer but the last got its sums from the layer above
    for a, r in zip(outs[True], outs[False]):
        s = max(float(r.abs().max()), 1e-3)
        assert float((a - r).abs().max()) <= 2e-5 * s


@pytest.mark.parametrize("rows,cin,widths,pool,relu_last", [
    (8192, 259, (256, 256, 512), 32, True),    # SA4 of the SSG model (materialised front end: odd row width, scalar loads)
    (16384, 320, (256, 128), 0, True),         # FP3
    (4096, 384, (256, 256), 0, True),          # FP2
    (32768, 128, (256,), 32, True),            # SA3's wide last layer + max over K
    (1000, 64, (128,), 0, False),              # ragged row count, one layer, no activation
    (96, 768, (256, 256), 0, True),            # FP1's widths (large first contraction)
    (64, 136, (512, 128, 256), 32, True),      # every slicing of the contraction (1, 4, 2 slices)
])
def test_mlp_wide_vs_float64(pn2, cuda, rows, cin, widths, pool, relu_last):
    """pn2_mlp_wide (a workgroup carries a 32-row tile through up to three 128/256/512-wide layers, weights streamed from
    L2, contraction slices added in LDS) against float64 of the same layers (pointnet_util.py:150-170 / :312-325 with the
    inference BN folded), and bit-for-bit run-to-run."""
    tfu = pn2.util.tf_util
    rs = np.random.RandomState(rows + cin)
    x = rs.randn(rows, cin).astype(np.float32)
    ws, bs, c = [], [], cin
    for w_ in widths:
        ws.append((rs.randn(c, w_) / np.sqrt(c)).astype(np.float32))
        bs.append((0.1 * rs.randn(w_)).astype(np.float32))
        c = w_
    y = tfu.hip_mlp_wide(T(x, cuda), [T(w_, cuda) for w_ in ws], [T(b_, cuda) for b_ in bs], relu_last=relu_last, pool=pool)
    assert y is not None
    h = x.astype(np.float64)
    for li, (w_, b_) in enumerate(zip(ws, bs)):
        h = h @ w_.astype(np.float64) + b_
        if li < len(ws) - 1 or relu_last:
            h = np.maximum(h, 0.0)
    if pool:
        h = h.reshape(rows // pool, pool, -1).max(1)
    assert y.shape == h.shape
    close(y.cpu().numpy(), h)
    y2 = tfu.hip_mlp_wide(T(x, cuda), [T(w_, cuda) for w_ in ws], [T(b_, cuda) for b_ in bs], relu_last=relu_last, pool=pool)
    assert (y2 == y).all()


@pytest.mark.parametrize("b,n,m,c,widths,pool", [(2, 256, 64, 256, (256, 256, 512), True), (3, 100, 10, 128, (128, 256), True),
                                                (1, 64, 16, 64, (512,), False)])
def test_sa_mlp_wide_vs_float64(pn2, cuda, b, n, m, c, widths, pool):
    """pn2_sa_mlp_wide: group_point + centre + concat [xyz | features] (pointnet_util.py:39-54) feeding the wide chain, max
    over the 32 neighbours; the weights keep the reference's row order."""
    tfu = pn2.util.tf_util
    rs = np.random.RandomState(n + c)
    xyz = rs.rand(b, n, 3).astype(np.float32)
    new_xyz = xyz[:, :m].copy()
    pts = rs.randn(b, n, c).astype(np.float32)
    idx = rs.randint(0, n, size=(b, m, 32)).astype(np.int32)
    ws, bs, cc = [], [], 3 + c
    for w_ in widths:
        ws.append((rs.randn(cc, w_) / np.sqrt(cc)).astype(np.float32))
        bs.append((0.1 * rs.randn(w_)).astype(np.float32))
        cc = w_
    kws = [tfu.sa_wide_first_layer(T(ws[0], cuda))] + [T(w_, cuda) for w_ in ws[1:]]
    y = tfu.hip_sa_mlp_wide(T(xyz, cuda), T(new_xyz, cuda), T(pts, cuda), T(idx, cuda), kws, [T(b_, cuda) for b_ in bs], pool=pool)
    assert y is not None
    bi = np.arange(b)[:, None, None]
    g = np.concatenate([xyz[bi, idx] - new_xyz[:, :, None, :], pts[bi, idx]], -1).astype(np.float64)  # (b,m,32,3+c)
    for w_, b_ in zip(ws, bs):
        g = np.maximum(g @ w_.astype(np.float64) + b_, 0.0)
    ref = g.max(2) if pool else g
    assert y.shape == ref.shape
    close(y.cpu().numpy(), ref)


@pytest.mark.parametrize("b,n,m,c1,c2,widths", [(2, 1024, 256, 64, 256, (256, 128)), (3, 64, 16, 0, 128, (128,)),
                                                 (1, 256, 64, 132, 60, (512, 256, 128))])
def test_fp_mlp_wide_vs_float64(pn2, cuda, b, n, m, c1, c2, widths):
    """pn2_fp_mlp_wide: inverse-distance weights + three_interpolate + concat [interpolated | points1]
    (pointnet_util.py:300-311) feeding the wide chain; against float64 and against the materialised front end
    (pn2_fp_interp_concat) + pn2_mlp_wide (bit-identical rows -> bit-identical result)."""
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    rs = np.random.RandomState(n + c2)
    xyz1 = rs.rand(b, n, 3).astype(np.float32)
    xyz2 = rs.rand(b, m, 3).astype(np.float32)
    dist, idx = pn2.tf_ops.tf_interpolate.three_nn(T(xyz1, cuda), T(xyz2, cuda))
    p2 = rs.randn(b, m, c2).astype(np.float32)
    p1 = rs.randn(b, n, c1).astype(np.float32) if c1 else None
    ws, bs, cc = [], [], c1 + c2
    for w_ in widths:
        ws.append((rs.randn(cc, w_) / np.sqrt(cc)).astype(np.float32))
        bs.append((0.1 * rs.randn(w_)).astype(np.float32))
        cc = w_
    tw, tb = [T(w_, cuda) for w_ in ws], [T(b_, cuda) for b_ in bs]
    tp1 = None if p1 is None else T(p1, cuda)
    y = tfu.hip_fp_mlp_wide(dist, idx, tp1, T(p2, cuda), tw, tb)
    assert y is not None and y.shape == (b * n, widths[-1])
    d = np.maximum(dist.cpu().numpy().astype(np.float64), 1e-10)
    w = (1.0 / d) / (1.0 / d).sum(2, keepdims=True)
    ii = idx.cpu().numpy()
    bi = np.arange(b)[:, None]
    h = sum(w[:, :, k, None] * p2.astype(np.float64)[bi, ii[:, :, k]] for k in range(3))
    if p1 is not None:
        h = np.concatenate([h, p1.astype(np.float64)], -1)
    h = h.reshape(b * n, -1)
    for w_, b_ in zip(ws, bs):
        h = np.maximum(h @ w_.astype(np.float64) + b_, 0.0)
    close(y.cpu().numpy(), h)
    x = pu._fp_interp_concat(dist, idx, tp1, T(p2, cuda), pad_to=8).reshape(b * n, -1)
    w0 = tw[0] if x.shape[1] == tw[0].shape[0] else __import__("torch").nn.functional.pad(tw[0], (0, 0, 0, x.shape[1] - tw[0].shape[0]))
    y2 = tfu.hip_mlp_wide(x, [w0.contiguous()] + tw[1:], tb)
    assert (y2 == y).all()


def test_mlp_wide_random_shapes_and_argument_checks(pn2, cuda):
    """20 random (rows, cin, widths, pool) draws of pn2_mlp_wide against float64 (ragged row counts, contraction lengths
    that are not multiples of 8 or 4, every width combination), then the entry points' refusals."""
    import ctypes
    tfu = pn2.util.tf_util
    rs = np.random.RandomState(7)
    for _ in range(20):
        nl = rs.randint(1, 4)
        widths = tuple(int(rs.choice([128, 256, 512])) for _ in range(nl))
        pool = 32 if rs.rand() < 0.4 else 0
        rows = int(rs.randint(1, 40)) * 32 if pool else int(rs.randint(1, 1500))
        cin = int(rs.randint(1, 300))
        x = rs.randn(rows, cin).astype(np.float32)
        ws, bs, c = [], [], cin
        for w_ in widths:
            ws.append((rs.randn(c, w_) / np.sqrt(c)).astype(np.float32))
            bs.append((0.1 * rs.randn(w_)).astype(np.float32))
            c = w_
        y = tfu.hip_mlp_wide(T(x, cuda), [T(w_, cuda) for w_ in ws], [T(b_, cuda) for b_ in bs], pool=pool)
        if y is None:  # does not fit LDS: a legitimate refusal only for wide inputs in front of a 512-wide second buffer
            assert cin > 256 or max(widths) == 512, (rows, cin, widths)
            continue
        h = x.astype(np.float64)
        for w_, b_ in zip(ws, bs):
            h = np.maximum(h @ w_.astype(np.float64) + b_, 0.0)
        if pool:
            h = h.reshape(rows // 32, 32, -1).max(1)
        close(y.cpu().numpy(), h)
    raw, L = pn2._lib._raw, pn2._lib
    x = T(np.zeros((64, 128), np.float32), cuda)
    w = T(np.zeros((128, 128), np.float32), cuda)
    b = T(np.zeros(128, np.float32), cuda)
    y = T(np.zeros((64, 128), np.float32), cuda)
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    wid = (ctypes.c_int * 1)(128)
    wid_bad = (ctypes.c_int * 1)(96)
    wp, bp = (ctypes.c_void_p * 1)(w.data_ptr()), (ctypes.c_void_p * 1)(b.data_ptr())
    call = lambda rows, cin, stride, nl, wd, pool: raw.pn2_mlp_wide(rows, cin, stride, P(x), nl, wd, wp, bp, 1, pool, P(y), None)  # noqa: E731
    assert call(64, 128, 128, 1, wid, 0) == 0
    assert call(64, 128, 128, 1, wid_bad, 0) == L.PN2_EUNSUP      # width not 128 / 256 / 512
    assert call(64, 128, 128, 4, wid, 0) == L.PN2_EUNSUP          # more than three layers
    assert call(64, 128, 128, 1, wid, 16) == L.PN2_EUNSUP         # pool other than 0 / 32
    assert call(70, 128, 128, 1, wid, 32) == -1         # pooled rows not a multiple of 32
    assert call(64, 128, 64, 1, wid, 0) == -1           # row stride shorter than the row
    assert call(0, 128, 128, 1, wid, 0) == -1
    assert raw.pn2_mlp_wide(64, 128, 128, None, 1, wid, wp, bp, 1, 0, P(y), None) == -2
