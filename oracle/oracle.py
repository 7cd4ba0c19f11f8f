"""numpy front-end of the CPU oracle (oracle/pn2_oracle.c) + numpy restatement of
the grouped shared-MLP maths.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.

Function names and argument orders follow the reference's Python op wrappers
(tf_ops/tf_sampling.py:38,61; tf_ops/tf_grouping.py:13,46; tf_ops/tf_interpolate.py:13,50).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpn2_oracle.so")

ARITH_STRICT, ARITH_FMA, ARITH_FMA_ALT = 0, 1, 2
# Defaults = what ONE complete build of the reference's kernels produces (oracle/_ref "fast_noslp": contraction on, as
# under nvcc's --fmad=true default): farthestpointsamplingKernel contracts as mode 2, query_ball_point_gpu as mode 1
# (read off the ISA, asserted by tests/test_ref_gpu.py).  The product's defaults (config.py) are the same pair.
DEFAULT_FPS_MODE = ARITH_FMA_ALT
DEFAULT_BQ_MODE = ARITH_FMA


def build(force=False):
    src = os.path.join(_HERE, "pn2_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpn2_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def num_threads():
    return int(lib().oracle_num_threads())


def set_num_threads(t):
    lib().oracle_set_num_threads(int(t))


def farthest_point_sample(npoint, inp, mode=DEFAULT_FPS_MODE):
    inp = _f32(inp)
    b, n, _ = inp.shape
    out = np.empty((b, npoint), dtype=np.int32)
    rc = lib().oracle_fps(b, n, int(npoint), _p(inp), _p(out), int(mode))
    assert rc == 0, rc
    return out


def fps_first_tie(npoint, inp, mode=DEFAULT_FPS_MODE):
    """first tied step of the FPS run (checker of pn2_fps_nested's tie record; see oracle_fps_first_tie) -> (b,) int32"""
    inp = _f32(inp)
    b, n, _ = inp.shape
    out = np.empty((b,), dtype=np.int32)
    rc = lib().oracle_fps_first_tie(b, n, int(npoint), _p(inp), _p(out), int(mode))
    assert rc == 0, rc
    return out


def gather_point(inp, idx):
    inp, idx = _f32(inp), _i32(idx)
    b, n, _ = inp.shape
    m = idx.shape[1]
    out = np.empty((b, m, 3), dtype=np.float32)
    lib().oracle_gather_point(b, n, m, _p(inp), _p(idx), _p(out))
    return out


def gather_point_grad(inp, idx, out_g):
    inp, idx, out_g = _f32(inp), _i32(idx), _f32(out_g)
    b, n, _ = inp.shape
    m = idx.shape[1]
    inp_g = np.empty((b, n, 3), dtype=np.float32)
    lib().oracle_gather_point_grad(b, n, m, _p(out_g), _p(idx), _p(inp_g))
    return inp_g


def query_ball_point(radius, nsample, xyz1, xyz2, mode=DEFAULT_BQ_MODE):
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = np.zeros((b, m, nsample), dtype=np.int32)  # zero-fill: documented divergence for empty balls
    cnt = np.empty((b, m), dtype=np.int32)
    rc = lib().oracle_query_ball_point(b, n, m, ctypes.c_float(radius), int(nsample), _p(xyz1), _p(xyz2),
                                       _p(idx), _p(cnt), int(mode))
    assert rc == 0, rc
    return idx, cnt


def group_point(points, idx):
    points, idx = _f32(points), _i32(idx)
    b, n, c = points.shape
    _, m, ns = idx.shape
    out = np.empty((b, m, ns, c), dtype=np.float32)
    lib().oracle_group_point(b, n, c, m, ns, _p(points), _p(idx), _p(out))
    return out


def group_point_grad(points, idx, grad_out):
    points, idx, grad_out = _f32(points), _i32(idx), _f32(grad_out)
    b, n, c = points.shape
    _, m, ns = idx.shape
    gp = np.empty((b, n, c), dtype=np.float32)
    lib().oracle_group_point_grad(b, n, c, m, ns, _p(grad_out), _p(idx), _p(gp))
    return gp


def three_nn(xyz1, xyz2):
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = np.empty((b, n, 3), dtype=np.float32)
    idx = np.empty((b, n, 3), dtype=np.int32)
    rc = lib().oracle_three_nn(b, n, m, _p(xyz1), _p(xyz2), _p(dist), _p(idx))
    assert rc == 0, rc
    return dist, idx


def three_interpolate(points, idx, weight):
    points, idx, weight = _f32(points), _i32(idx), _f32(weight)
    b, m, c = points.shape
    n = idx.shape[1]
    out = np.empty((b, n, c), dtype=np.float32)
    lib().oracle_three_interpolate(b, m, c, n, _p(points), _p(idx), _p(weight), _p(out))
    return out


def three_interpolate_grad(points, idx, weight, grad_out):
    points, idx, weight, grad_out = _f32(points), _i32(idx), _f32(weight), _f32(grad_out)
    b, m, c = points.shape
    n = idx.shape[1]
    gp = np.empty((b, m, c), dtype=np.float32)
    lib().oracle_three_interpolate_grad(b, n, c, m, _p(grad_out), _p(idx), _p(weight), _p(gp))
    return gp


def select_top_k(k, dist):
    """tf_ops/tf_grouping.py:31-40 -> (idx (b,m,n) int32, dist_out (b,m,n)): first k sorted, rest swapped."""
    dist = _f32(dist)
    b, m, n = dist.shape
    outi = np.empty((b, m, n), dtype=np.int32)
    out = np.empty((b, m, n), dtype=np.float32)
    rc = lib().oracle_selection_sort(b, n, m, int(k), _p(dist), _p(outi), _p(out))
    assert rc == 0, rc
    return outi, out


def knn_point(k, xyz1, xyz2):
    """tf_ops/tf_grouping.py:64-89: dist = reduce_sum((xyz1 - xyz2)^2, -1) in fp32 (separate TF ops:
    no contraction; summed x^2 + y^2 + z^2 left to right), then select_top_k -> (val (b,m,k), idx (b,m,k))."""
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    d = (xyz1[:, None, :, :] - xyz2[:, :, None, :]) ** np.float32(2)
    dist = ((d[..., 0] + d[..., 1]) + d[..., 2]).astype(np.float32)
    outi, out = select_top_k(k, dist)
    return out[:, :, :k], outi[:, :, :k]


def prob_sample(inp, inpr):
    """tf_ops/tf_sampling.py:18-26 -> (out (b,m) int32, cumsum (b,n) float32 as the reference's scratch holds it)."""
    inp, inpr = _f32(inp), _f32(inpr)
    b, n = inp.shape
    m = inpr.shape[1]
    temp = np.empty((b, n), dtype=np.float32)
    out = np.empty((b, m), dtype=np.int32)
    rc = lib().oracle_prob_sample(b, n, m, _p(inp), _p(inpr), _p(temp), _p(out))
    assert rc == 0, rc
    return out, temp


def interpolate_label_with_color(sparse_points, sparse_labels, dense_points, knn):
    """tf_ops/tf_interpolate.py:28-44 -> (dense_labels (Nd,) int32, dense_colors (Nd,3) uint8)."""
    sp, dp = _f32(sparse_points), _f32(dense_points)
    sl = _i32(sparse_labels)
    ns, nd = sp.shape[0], dp.shape[0]
    labels = np.empty((nd,), dtype=np.int32)
    colors = np.empty((nd, 3), dtype=np.uint8)
    rc = lib().oracle_interpolate_label_with_color(ns, nd, _p(sp), _p(sl), _p(dp), _p(labels), _p(colors), int(knn))
    assert rc == 0, rc
    return labels, colors


# ---------------------------------------------------------------------------
# Layer maths (numpy).  These restate util/pointnet_util.py + util/tf_util.py.
# `dtype` float64 gives the high-precision oracle the 1e-5 feature tolerance is
# measured against; float32 gives a same-precision CPU baseline.
# ---------------------------------------------------------------------------
BN_EPS = 1e-3  # tf.contrib.layers.batch_norm default epsilon (util/tf_util.py:571-581)


def fp_weights(dist):
    """util/pointnet_util.py:300-303 -- fp32 elementwise, as TF would run it."""
    dist = np.maximum(_f32(dist), np.float32(1e-10))
    inv = np.float32(1.0) / dist
    norm = np.sum(inv, axis=2, keepdims=True, dtype=np.float32)
    return (inv / norm).astype(np.float32)


def conv_bn_relu(x, layer, dtype=np.float64, relu=True):
    """1x1 conv (= matmul over the last dim) + bias + inference BN + ReLU.
    util/tf_util.py:181-203 (conv2d), :571-581 (batch_norm, is_training=False):
    y = (x@W + b - mean) / sqrt(var + 1e-3) * gamma + beta.
    layer: dict(W (Cin,Cout), b, gamma, beta, mean, var) or with bn=False only W,b."""
    W = np.asarray(layer["W"], dtype=dtype)
    y = x.astype(dtype) @ W + np.asarray(layer["b"], dtype=dtype)
    if layer.get("gamma") is not None:
        g = np.asarray(layer["gamma"], dtype=dtype)
        be = np.asarray(layer["beta"], dtype=dtype)
        mu = np.asarray(layer["mean"], dtype=dtype)
        var = np.asarray(layer["var"], dtype=dtype)
        y = (y - mu) / np.sqrt(var + dtype(BN_EPS)) * g + be
    if relu:
        y = np.maximum(y, 0)
    return y


def sample_and_group(npoint, radius, nsample, xyz, points, use_xyz=True, mode=None):
    """util/pointnet_util.py:18-60 (knn=False)."""
    fidx = farthest_point_sample(npoint, xyz, DEFAULT_FPS_MODE if mode is None else mode)  # mode: ONE value for both ops
    new_xyz = gather_point(xyz, fidx)
    idx, _ = query_ball_point(radius, nsample, xyz, new_xyz, DEFAULT_BQ_MODE if mode is None else mode)
    grouped_xyz = group_point(xyz, idx) - new_xyz[:, :, None, :]
    if points is not None:
        gp = group_point(points, idx)
        new_points = np.concatenate([grouped_xyz, gp], axis=-1) if use_xyz else gp
    else:
        new_points = grouped_xyz
    return new_xyz, new_points, idx, grouped_xyz


def sa_module(xyz, points, npoint, radius, nsample, layers, dtype=np.float64, mode=None, layers2=None, use_xyz=True,
              group_all=False):
    """pointnet_sa_module, pooling='max', is_training=False (util/pointnet_util.py:98-216).  layers2: the `mlp2` stack
    applied to the pooled (B,npoint,1,C) tensor (:194-211); use_xyz=False: grouped features only (:52-58);
    group_all=True: one group of all points around the origin, [xyz, features] (:63-95,137-141)."""
    if group_all:
        b, n, _ = xyz.shape
        new_xyz = np.zeros((b, 1, 3), np.float32)                       # :81
        idx = np.tile(np.arange(n, dtype=np.int32).reshape(1, 1, n), (b, 1, 1))  # :82
        gx = _f32(xyz).reshape(b, 1, n, 3)                              # :83
        if points is not None:
            new_points = np.concatenate([gx, _f32(points)[:, None]], axis=-1) if use_xyz else _f32(points)[:, None]  # :85-90
        else:
            new_points = gx
    else:
        new_xyz, new_points, idx, _ = sample_and_group(npoint, radius, nsample, xyz, points, use_xyz, mode)
    h = new_points
    for layer in layers:
        h = conv_bn_relu(h, layer, dtype)
    h = h.max(axis=2, keepdims=True)
    for layer in (layers2 or []):
        h = conv_bn_relu(h, layer, dtype)
    return new_xyz, h[:, :, 0], idx


def fp_module(xyz1, xyz2, points1, points2, layers, dtype=np.float64):
    """pointnet_fp_module, is_training=False (util/pointnet_util.py:285-326)."""
    dist, idx = three_nn(xyz1, xyz2)
    w = fp_weights(dist)
    interp = three_interpolate(points2, idx, w)
    h = np.concatenate([interp, points1], axis=2) if points1 is not None else interp
    for layer in layers:
        h = conv_bn_relu(h, layer, dtype)
    return h


def bf16_round(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32 -- what v_cvt_pk_bf16_f32 does."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


def mlp_max_bf16(grouped_xyz, grouped_points_bf16, ws, bs):
    """The precision contract of pn2_sa_mlp_max_fused_bf16 (csrc/pn2_sa_fused_bf16.hip), restated:
    input = [bf16(grouped_xyz) | features already bf16]; per layer W is rounded to bf16, products are exact,
    the sum is taken here in float64 (the kernel: fp32 MFMA accumulation), + fp32 bias, ReLU; hidden
    activations are rounded to bf16; the result is the max over the K neighbours of the last layer.
    grouped_xyz (B,M,K,3) fp32 = xyz[idx] - new_xyz, grouped_points_bf16 (B,M,K,C) fp32 holding bf16 values,
    ws[l] (cin_l, w_l) fp32 (BN already folded), bs[l] (w_l,) fp32."""
    h = np.concatenate([bf16_round(grouped_xyz), _f32(grouped_points_bf16)], axis=-1).astype(np.float64)
    for l, (w, b) in enumerate(zip(ws, bs)):
        y = h @ bf16_round(w).astype(np.float64) + np.asarray(b, np.float64)
        y = np.maximum(y, 0.0)
        if l + 1 < len(ws):
            h = bf16_round(y.astype(np.float32)).astype(np.float64)
    return y.max(axis=2)


def model_head(feats, fc1, fc2, dtype=np.float64):
    """model.py:131-146 at inference: conv1d(128)+BN+ReLU -> dropout (identity when not training,
    util/tf_util.py:587-606) -> conv1d(num_class), no activation, no BN."""
    h = conv_bn_relu(feats, fc1, dtype)
    return conv_bn_relu(h, fc2, dtype, relu=False)


def weighted_sparse_ce(pred, label, smpw):
    """model.py:152-161: tf.losses.sparse_softmax_cross_entropy(labels, logits, weights) with the default
    reduction SUM_BY_NONZERO_WEIGHTS = sum(w_i * ce_i) / #{i : w_i != 0} (0 when every weight is 0)."""
    z = np.asarray(pred, np.float64).reshape(-1, pred.shape[-1])
    y = np.asarray(label).reshape(-1).astype(np.int64)
    w = np.asarray(smpw, np.float64).reshape(-1)
    z = z - z.max(axis=1, keepdims=True)
    ce = np.log(np.exp(z).sum(axis=1)) - z[np.arange(z.shape[0]), y]
    nz = np.count_nonzero(w)
    return float((ce * w).sum() / nz) if nz else 0.0


BN_EPSILON = 1e-3  # tf.contrib.layers.batch_norm default epsilon (util/tf_util.py:571-581 passes none)


def batch_norm_relu_train(y, gamma, beta, relu=True, eps=BN_EPSILON, bias=None, moving=None, decay=0.9):
    """Training-mode tf.contrib.layers.batch_norm + tf.nn.relu on a (rows, c) layer output
    (util/tf_util.py:555-581, called from conv2d / conv1d at :186-204): batch moments over the rows (biased
    variance), z = relu(gamma*(y-mean)/sqrt(var+eps)+beta).  Returns (z, mean, var[, new_moving_mean,
    new_moving_var]); the moving averages follow tf's fused kernel: mean of the layer output INCLUDING `bias`
    (the per-channel constant added before BN) and the unbiased batch variance, blended with `decay`."""
    y = np.asarray(y, np.float64)
    mean = y.mean(axis=0)
    var = y.var(axis=0)
    z = (y - mean) / np.sqrt(var + eps) * np.asarray(gamma, np.float64) + np.asarray(beta, np.float64)
    if relu:
        z = np.maximum(z, 0.0)
    if moving is None:
        return z, mean, var
    rows = y.shape[0]
    mm, mv = (np.asarray(a, np.float64) for a in moving)
    m_out = mean + (0.0 if bias is None else np.asarray(bias, np.float64))
    var_unb = var * rows / (rows - 1) if rows > 1 else var
    return z, mean, var, decay * mm + (1.0 - decay) * m_out, decay * mv + (1.0 - decay) * var_unb


def batch_norm_relu_train_grad(y, gamma, beta, dz, relu=True, eps=BN_EPSILON, mask=None):
    """Gradient of batch_norm_relu_train w.r.t. (y, gamma, beta), as tf.gradients derives it:
    g = dz*[z>0], xhat = (y-mean)/sqrt(var+eps), dbeta = sum g, dgamma = sum g*xhat,
    dy = gamma/sqrt(var+eps) * (g - mean(g) - xhat*mean(g*xhat)).  `mask` overrides [z>0] (to use the mask a
    float32 implementation took for elements within rounding of zero)."""
    y = np.asarray(y, np.float64)
    gamma = np.asarray(gamma, np.float64)
    mean = y.mean(axis=0)
    invstd = 1.0 / np.sqrt(y.var(axis=0) + eps)
    xhat = (y - mean) * invstd
    g = np.asarray(dz, np.float64)
    if relu:
        if mask is None:
            mask = (xhat * gamma + np.asarray(beta, np.float64)) > 0
        g = g * mask
    dbeta = g.sum(axis=0)
    dgamma = (g * xhat).sum(axis=0)
    dy = gamma * invstd * (g - g.mean(axis=0) - xhat * (g * xhat).mean(axis=0))
    return dy, dgamma, dbeta


def max_pool_rows(z, pool):
    """tf.reduce_max over the K neighbours of an SA layer (util/pointnet_util.py:167-170) on a (rows, c) activation
    whose rows come in consecutive groups of `pool`: -> (zmax (rows/pool, c), ties (rows/pool, c) = how many rows of
    the group attain the maximum)."""
    z = np.asarray(z, np.float64)
    g = z.reshape(z.shape[0] // pool, pool, z.shape[1])
    zmax = g.max(axis=1)
    return zmax, (g == zmax[:, None, :]).sum(axis=1).astype(np.float64)


def max_pool_rows_grad(z, pool, dzmax):
    """Gradient of max_pool_rows w.r.t. z as tf.gradients / torch.amax define it: the incoming gradient is shared
    equally among the rows that attain the maximum."""
    z = np.asarray(z, np.float64)
    g = z.reshape(z.shape[0] // pool, pool, z.shape[1])
    sel = g == g.max(axis=1, keepdims=True)
    return (sel * (np.asarray(dzmax, np.float64) / sel.sum(axis=1))[:, None, :]).reshape(z.shape)
