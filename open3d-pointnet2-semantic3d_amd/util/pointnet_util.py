"""PointNet++ layers with the reference's API on the MI355X kernels.

Keeps the names, argument orders and return values of util/pointnet_util.py:
sample_and_group (:18-60), sample_and_group_all (:63-95), pointnet_sa_module
(:98-216), pointnet_sa_module_msg (:219-282), pointnet_fp_module (:285-326).
Tensors are torch float32/int32 CUDA(ROCm) tensors; variables live in the
VariableStore of util/tf_util.py under the same scope names the reference uses
("layer1/conv0/weights", "fa_layer4/conv_2/bn/gamma", ...).

Inference (is_training=False) never materialises the grouped (B,M,K,C) tensor
when the fused kernel applies; training uses the HIP index/gather ops with
autograd plus differentiable torch layers.
"""
import ctypes

import torch

from . import tf_util
from .._lib import PN2_EUNSUP, check, lib, ptr, require_cuda, rows_in_place, stream_ptr
from ..tf_ops import tf_grouping
from ..tf_ops.tf_grouping import query_ball_point_multi, group_point, knn_point, query_ball_point
from ..tf_ops.tf_interpolate import three_interpolate, three_nn
from ..tf_ops.tf_sampling import (farthest_point_sample, farthest_point_sample_and_gather, farthest_point_sample_with_ties,
                                  gather_point, tag_fps_output)


def sample_and_group(npoint, radius, nsample, xyz, points, knn=False, use_xyz=True, geometry=None):
    """-> new_xyz (B,npoint,3), new_points (B,npoint,nsample,3+C), idx, grouped_xyz.
    geometry = (new_xyz, idx[, plan]) (extension): the weight-independent half -- FPS, gather, ball query, and optionally
    the scatter plan of the grouping's gradient (scatter_plan) -- was computed ahead (model.compute_geometry on a side
    stream); only the grouping runs here."""
    plan = None
    if geometry is not None:
        new_xyz, idx = geometry[0], geometry[1]
        plan = geometry[2] if len(geometry) > 2 else None
    else:
        fps_idx, fps_tie, _ = farthest_point_sample_with_ties(npoint, xyz)  # (:36-37; the tie record rides on new_xyz, see
        new_xyz = tag_fps_output(gather_point(xyz, fps_idx), fps_tie)         #  tf_sampling.USE_NESTED_FPS)
        if knn:
            _, idx = knn_point(nsample, xyz, new_xyz)
        else:
            idx, pts_cnt = query_ball_point(radius, nsample, xyz, new_xyz)
    if USE_FUSED_TRAIN_FRONT and points is not None and use_xyz and points.dtype == torch.float32:
        # gather + centre + concat in one launch; grouped_xyz is a view of its first three columns.  Features that carry no
        # gradient (the colours of the level-0 module) need no tape node: the same launch, called directly
        if points.requires_grad and torch.is_grad_enabled():
            new_points = _SAGroupConcat.apply(xyz.contiguous(), new_xyz.contiguous(), points.contiguous(), idx, plan)
        else:
            new_points = _sa_group_concat(xyz.detach().contiguous(), new_xyz.detach().contiguous(), points.detach(), idx)
        return new_xyz, new_points, idx, new_points[..., :3]
    grouped_xyz = group_point(xyz, idx)
    grouped_xyz = grouped_xyz - new_xyz.unsqueeze(2)  # translation normalisation (:44-46)
    if points is not None:
        grouped_points = group_point(points, idx)
        if use_xyz:
            new_points = torch.cat([grouped_xyz, grouped_points], dim=-1)  # xyz FIRST (:52-54)
        else:
            new_points = grouped_points
    else:
        new_points = grouped_xyz
    return new_xyz, new_points, idx, grouped_xyz


def sample_and_group_all(xyz, points, use_xyz=True):
    """npoint=1, radius=inf, centroid (0,0,0) (:63-95)."""
    b, n, _ = xyz.shape
    new_xyz = torch.zeros((b, 1, 3), dtype=torch.float32, device=xyz.device)
    idx = torch.arange(n, dtype=torch.int32, device=xyz.device).reshape(1, 1, n).repeat(b, 1, 1)
    grouped_xyz = xyz.reshape(b, 1, n, 3)
    if points is not None:
        new_points = torch.cat([xyz, points], dim=2) if use_xyz else points
        new_points = new_points.unsqueeze(1)
    else:
        new_points = grouped_xyz
    return new_xyz, new_points, idx, grouped_xyz


def _sa_fused_inference(xyz, new_xyz, points, idx, mlp, bn, conv_scope_fmt, pool=True, xyz_last=False):
    """Try the fully fused gather+MLP(+max) kernel; returns None if the configuration is outside what
    pn2_sa_mlp_max_fused / pn2_sa_mlp_rows_fused (pool=False: un-pooled (B,M,K,w) output) support.
    xyz_last: the layer's variables expect [features | xyz] (the MSG module's order); the kernel always feeds
    [xyz | features], so the first layer's weight rows are rotated accordingly (same products, same sum)."""
    b, n, _ = xyz.shape
    m, nsample = idx.shape[1], idx.shape[2]
    c = 0 if points is None else points.shape[2]
    bf16 = points is not None and points.dtype == torch.bfloat16  # BASELINE configs[4]: bf16 features, K = 32 or 64
    if bf16:
        if not pool or nsample not in (32, 64) or c % 16 != 0:
            return None
    elif nsample != 32 and (not pool or nsample not in (16, 64, 128, 256)):
        return None
    if len(mlp) > 3 or any(w % 32 != 0 or w > 128 for w in mlp):
        return None
    ws, bs = [], []
    cin = 3 + c
    for i, cout in enumerate(mlp):
        with tf_util.variable_scope(conv_scope_fmt % i):
            w2, b2 = tf_util.folded_dense(cin, cout, bn, (1, 1, cin, cout),
                                          rotate_rows=3 if (xyz_last and i == 0 and c > 0) else 0)
        ws.append(w2)
        bs.append(b2)
        cin = cout
    L = len(mlp)
    widths = (ctypes.c_int * L)(*mlp)
    oshape = (b, m, mlp[-1]) if pool else (b, m, nsample, mlp[-1])
    out = torch.empty(oshape, dtype=torch.float32, device=xyz.device)
    if (USE_HOISTED_SA and not bf16 and nsample == 32 and c >= 32 and c % 4 == 0
            and ((pool and tuple(mlp) in ((64, 64, 128), (128, 128, 128))) or (not pool and tuple(mlp) == (128, 128)))):
        # feature part of the first layer hoisted by linearity: zf = points @ W1[3:] on the n source points (8x fewer rows
        # than the m*K grouped neighbours at every level of semantic.json), its rows gathered into the accumulators
        w1x, w1f = tf_util.split_first_layer(ws[0], 3, c, "sa_pre")   # rows [0,3) = xyz, [3, 3+c) = features
        xyz, points = xyz.contiguous(), points.contiguous()
        zf = tf_util.hoist_gemm(points.reshape(b * n, c), w1f)
        wl = [w1x] + ws[1:]
        wptrs = (ctypes.c_void_p * L)(*[t.data_ptr() for t in wl])
        bptrs = (ctypes.c_void_p * L)(*[t.data_ptr() for t in bs])
        with torch.cuda.device(xyz.device):
            rc = lib.pn2_sa_mlp_fused_pre(b, n, m, nsample, ptr(xyz), ptr(new_xyz), ptr(zf), ptr(idx), L,
                                          ctypes.cast(widths, ctypes.c_void_p), ctypes.cast(wptrs, ctypes.c_void_p),
                                          ctypes.cast(bptrs, ctypes.c_void_p), int(bool(pool)), ptr(out), stream_ptr())
        if rc != PN2_EUNSUP:
            check(rc, "pn2_sa_mlp_fused_pre")
            return out
    wptrs = (ctypes.c_void_p * L)(*[t.data_ptr() for t in ws])
    bptrs = (ctypes.c_void_p * L)(*[t.data_ptr() for t in bs])
    xyz_v, ldx = rows_in_place(xyz)
    pts, ldp = (None, 0) if points is None else rows_in_place(points)
    if pool and not bf16 and (ldx != 3 or (points is not None and ldp != c)):
        # column blocks of a wider batch (model.get_sa_fp_features: the xyz / rgb halves of point_cloud (b,n,6)) gathered in place
        with torch.cuda.device(xyz.device):
            rc = lib.pn2_sa_mlp_max_fused_ld(b, n, m, nsample, c, ptr(xyz_v), ldx, ptr(new_xyz), ptr(pts), ldp, ptr(idx), L,
                                             ctypes.cast(widths, ctypes.c_void_p), ctypes.cast(wptrs, ctypes.c_void_p),
                                             ctypes.cast(bptrs, ctypes.c_void_p), ptr(out), stream_ptr())
        if rc != PN2_EUNSUP:
            check(rc, "pn2_sa_mlp_max_fused_ld")
            return out
    xyz_v = xyz_v.contiguous()
    pts = None if pts is None else pts.contiguous()
    fn = lib.pn2_sa_mlp_max_fused_bf16 if bf16 else (lib.pn2_sa_mlp_max_fused if pool else lib.pn2_sa_mlp_rows_fused)
    with torch.cuda.device(xyz.device):
        rc = fn(b, n, m, nsample, c, ptr(xyz_v), ptr(new_xyz), ptr(pts), ptr(idx), L,
                                      ctypes.cast(widths, ctypes.c_void_p), ctypes.cast(wptrs, ctypes.c_void_p),
                                      ctypes.cast(bptrs, ctypes.c_void_p), ptr(out), stream_ptr())
    if rc == PN2_EUNSUP:
        return None
    check(rc, "pn2_sa_mlp_max_fused")
    return out


def _sa_group_concat(xyz, new_xyz, points, idx):
    b, n, _ = xyz.shape
    m, nsample = idx.shape[1], idx.shape[2]
    c = 0 if points is None else points.shape[2]
    out = torch.empty((b, m, nsample, 3 + c), dtype=torch.float32, device=xyz.device)
    pts = None if points is None else points.contiguous()
    with torch.cuda.device(xyz.device):
        check(lib.pn2_sa_group_concat(b, n, m, nsample, c, ptr(xyz), ptr(new_xyz), ptr(pts), ptr(idx), ptr(out),
                                      stream_ptr()), "pn2_sa_group_concat")
    return out



def scatter_plan(idx, nsrc, weight=None, weight_kind=None):
    """The (weight-independent) list behind the gradients of group_point / three_interpolate, built ahead of the backward
    pass (pn2_scatter_plan_build): idx (b, rows, k) int32 into nsrc source points; weight (b, rows, k) or None.
    weight_kind 2: `weight` holds three_nn's squared distances.  -> opaque uint8 tensor."""
    b = idx.shape[0]
    nent = idx.shape[1] * idx.shape[2]
    div = idx.shape[2] if weight is not None else 1
    kind = (1 if weight_kind is None else weight_kind) if weight is not None else 0
    nbytes = lib.pn2_scatter_plan_bytes(b, nent, nsrc)
    plan = torch.empty(nbytes, dtype=torch.uint8, device=idx.device)
    with torch.cuda.device(idx.device):
        check(lib.pn2_scatter_plan_build(b, nent, div, nsrc, ptr(idx), ptr(weight), kind, ptr(plan), nbytes, stream_ptr()),
              "pn2_scatter_plan_build")
    return plan


def scatter_plans(specs):
    """several scatter plans of one batch in ONE memset + three launches (pn2_scatter_plan_build_multi) instead of four launches
    each: specs = [(idx (b, rows, k) int32, nsrc, weight or None, weight_kind or None), ...] as scatter_plan takes them (None
    entries pass through) -> list of plans, slices of one buffer.  The geometry stream of a training step builds its seven
    plans this way (model.compute_geometry)."""
    live = [(i, sp) for i, sp in enumerate(specs) if sp is not None]
    out = [None] * len(specs)
    for lo in range(0, len(live), 8):
        chunk = live[lo:lo + 8]
        b = chunk[0][1][0].shape[0]
        dev = chunk[0][1][0].device
        nent, div, nsrc, kinds, idxs, ws, offs, total = [], [], [], [], [], [], [], 0
        for _, (idx, ns, weight, weight_kind) in chunk:
            if idx.shape[0] != b:
                raise ValueError("scatter_plans: one batch size")
            ne = idx.shape[1] * idx.shape[2]
            nent.append(ne)
            div.append(idx.shape[2] if weight is not None else 1)
            nsrc.append(int(ns))
            kinds.append((1 if weight_kind is None else weight_kind) if weight is not None else 0)
            idxs.append(idx.contiguous())
            ws.append(None if weight is None else weight.contiguous())
            offs.append(total)
            total += (lib.pn2_scatter_plan_bytes(b, ne, int(ns)) + 15) // 16 * 16
        buf = torch.empty(total, dtype=torch.uint8, device=dev)
        k = len(chunk)
        ia = lambda v: (ctypes.c_int * k)(*v)  # noqa: E731
        pa = lambda ts: (ctypes.c_void_p * k)(*[None if t is None else t.data_ptr() for t in ts])  # noqa: E731
        with torch.cuda.device(dev):
            check(lib.pn2_scatter_plan_build_multi(k, b, ia(nent), ia(div), ia(nsrc), pa(idxs), pa(ws), ia(kinds), ptr(buf),
                                                   (ctypes.c_size_t * k)(*offs), total, stream_ptr()), "pn2_scatter_plan_build_multi")
        for j, (i, _) in enumerate(chunk):
            out[i] = buf[offs[j]:offs[j] + lib.pn2_scatter_plan_bytes(b, nent[j], nsrc[j])]
    return out


def _scatter_plan_apply(plan, rows_in, col0, c, nent, div, nsrc):
    """columns [col0, col0+c) of rows_in (b, ..., width) (read in place) scattered through `plan` -> (b, nsrc, c)"""
    b, width = rows_in.shape[0], rows_in.shape[-1]
    if not rows_in.is_contiguous():
        rows_in = rows_in.contiguous()
    out = torch.empty((b, nsrc, c), dtype=torch.float32, device=rows_in.device)
    with torch.cuda.device(rows_in.device):
        check(lib.pn2_scatter_plan_apply(b, nent, div, c, nsrc, ctypes.c_void_p(rows_in.data_ptr() + 4 * col0), width,
                                         ptr(plan), plan.numel(), ptr(out), stream_ptr()), "pn2_scatter_plan_apply")
    return out


def _plan_usable(plan, c):
    return plan is not None and c % 4 == 0 and c <= 1024


class _SAGroupConcat(torch.autograd.Function):
    """[group_point(xyz) - new_xyz | group_point(points)] in ONE launch (pn2_sa_group_concat) for the training path
    (util/pointnet_util.py:39-54 as four TF ops: two gathers, tile + subtract, concat).  Gradient w.r.t. `points` only
    (coordinates are data): the feature columns of the upstream gradient go through the list-and-gather
    pn2_group_point_grad_ws -- or, with a `plan` built ahead (scatter_plan(idx, n)), through the gather alone, reading the
    feature columns of the upstream gradient in place."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, points, idx, plan=None):
        out = _sa_group_concat(xyz, new_xyz, points, idx)
        ctx.save_for_backward(idx, plan)
        ctx.n, ctx.c = xyz.shape[1], points.shape[2]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, plan = ctx.saved_tensors
        b, m, ns, _ = grad_out.shape
        if _plan_usable(plan, ctx.c):
            return None, None, _scatter_plan_apply(plan, grad_out, 3, ctx.c, m * ns, 1, ctx.n), None, None
        g = grad_out[..., 3:].contiguous()  # (b,m,ns,c)
        gp = torch.empty((b, ctx.n, ctx.c), dtype=torch.float32, device=g.device)
        nbytes = lib.pn2_group_point_grad_workspace_bytes(b, ctx.n, m, ns)
        ws = torch.empty(nbytes // 4, dtype=torch.int32, device=g.device)
        with torch.cuda.device(g.device):
            check(lib.pn2_group_point_grad_ws(b, ctx.n, ctx.c, m, ns, ptr(g), ptr(idx), ptr(gp), ptr(ws), nbytes,
                                              stream_ptr()), "pn2_group_point_grad_ws")
        return None, None, gp, None, None


class _FPInterpConcat(torch.autograd.Function):
    """[three_interpolate(points2, idx, w(dist)) | points1] in ONE launch (pn2_fp_interp_concat) for the training path
    (util/pointnet_util.py:300-311: clamp, reciprocal, sum, divide, three_interpolate, concat).  Gradients: points2
    through pn2_three_interpolate_grad_ws with the same weights -- or the gather of a `plan` built ahead
    (scatter_plan(idx, m, dist, weight_kind=2)) -- points1 = its slice of the upstream gradient."""

    @staticmethod
    def forward(ctx, dist, idx, points1, points2, plan=None):
        out = _fp_interp_concat(dist, idx, points1, points2)
        ctx.save_for_backward(dist, idx, plan)
        ctx.m, ctx.c2 = points2.shape[1], points2.shape[2]
        ctx.c1 = 0 if points1 is None else points1.shape[2]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        dist, idx, plan = ctx.saved_tensors
        g1 = grad_out[..., ctx.c2:] if ctx.c1 and ctx.needs_input_grad[2] else None
        if not ctx.needs_input_grad[3]:
            return None, None, g1, None, None
        if _plan_usable(plan, ctx.c2):
            n = idx.shape[1]
            return None, None, g1, _scatter_plan_apply(plan, grad_out, 0, ctx.c2, 3 * n, 3, ctx.m), None
        d = torch.clamp(dist, min=1e-10)
        w = (1.0 / d)
        w = (w / w.sum(dim=2, keepdim=True)).contiguous()
        from ..tf_ops.tf_interpolate import three_interpolate_grad
        g2 = three_interpolate_grad(grad_out[..., :ctx.c2].contiguous(), idx, w, ctx.m)
        return None, None, g1, g2, None


# set False to force the unfused HIP path (group_concat + pn2_linear); used by tests/bench
USE_FUSED_SA = True


USE_HOISTED_SA = True  # A/B: feature part of the first SA layer computed on the source points (linearity)
USE_HOISTED_FP = True  # A/B: first FP layer's product with the interpolated channels computed on the known points (linearity)
def sa_geometry(xyz, npoint, radius, nsample):
    """The feature-independent half of an SA layer: FPS -> gather -> ball query.
    -> new_xyz (B,npoint,3), idx (B,npoint,nsample).  Depends only on coordinates, so a model can
    run the geometry of all levels on a side stream (see model.get_sa_fp_features).
    (The binned-once ball query -- tf_grouping.ball_query_bin + query_ball_point_binned -- pays when the binning sits in the
    sampler half of a pipelined batch: model.sa1_samples; forked onto a second stream inside one batch's graph it cost the
    throughput regime 0.49 -> 0.64 ms per step, r03, and was removed from this function in r06.)"""
    _, new_xyz = farthest_point_sample_and_gather(npoint, xyz)  # one launch: the FPS kernel emits the coordinates
    idx, _ = query_ball_point(radius, nsample, xyz, new_xyz)
    return new_xyz, idx


# ---- the coarse levels in one launch (pn2_coarse_geometry) ----------------------------------------------------------------
# Below the first SA level the clouds are tiny (semantic.json: 1024 -> 256 -> 64 -> 16 points): FPS (or its nested shortcut),
# ball query and the FP side's three_nn of a level are three launches of 6-13 us for a few us of work, all on the critical
# path of a batch.  They depend on coordinates only, so one workgroup per cloud walks down the levels in ONE launch; every
# output is the bits of the separate ops.  False = the separate ops (A/B, tests).
USE_COARSE_GEOMETRY = True
COARSE_MAX_N, COARSE_MAX_LEVELS, COARSE_MAX_NN_M = 1024, 4, 256


def coarse_geometry_fits(n0, npoints, want_nn=True):
    """can pn2_coarse_geometry take levels of `npoints` samples under a source cloud of n0 points?"""
    if not USE_COARSE_GEOMETRY or not (0 < n0 <= COARSE_MAX_N) or not (0 < len(npoints) <= COARSE_MAX_LEVELS):
        return False
    n = n0
    for m in npoints:
        if not (0 < m <= n) or (want_nn and not (3 <= m <= COARSE_MAX_NN_M)):
            return False
        n = m
    return True


def coarse_geometry(xyz0, npoints, radii, nsamples, want_nn=True, fps_arith_mode=None, bq_arith_mode=None):
    """Levels l = 0 .. len(npoints)-1 under xyz0 (b,n0,3), the source cloud of level l being the samples of level l-1:
    -> [{"new_xyz" (b,m,3), "fps_idx" (b,m), "idx" (b,m,nsample), "cnt" (b,m), "nn": (dist, idx) (b,n,3) or None}] --
    farthest_point_sample + gather_point, query_ball_point and (want_nn) three_nn(source cloud, new_xyz) of every level,
    bit for bit, in one launch (pn2_coarse_geometry).  xyz0 carrying the tie record of the FPS run that produced it
    (tf_sampling.tag_fps_output) lets the clouds without ties skip the sampling, as farthest_point_sample does."""
    from .. import config
    from ..tf_ops.tf_sampling import fps_tie_record
    require_cuda(xyz0)
    if xyz0.dim() != 3 or xyz0.shape[2] != 3 or xyz0.dtype != torch.float32:
        raise ValueError("coarse_geometry expects (batch_size,num_points,3) float32 xyz0")
    nlev = len(npoints)
    if not (len(radii) == len(nsamples) == nlev):
        raise ValueError("coarse_geometry expects one radius and one nsample per level")
    tie_in = fps_tie_record(xyz0, fps_arith_mode)
    xyz0 = xyz0.detach().contiguous()
    b, n0, _ = xyz0.shape
    dev = xyz0.device
    out, n = [], n0
    for m, ns in zip(npoints, nsamples):
        lv = {"new_xyz": torch.empty((b, m, 3), dtype=torch.float32, device=dev),
              "fps_idx": torch.empty((b, m), dtype=torch.int32, device=dev),
              "idx": torch.empty((b, m, ns), dtype=torch.int32, device=dev),
              "cnt": torch.empty((b, m), dtype=torch.int32, device=dev),
              "nn": (torch.empty((b, n, 3), dtype=torch.float32, device=dev),
                     torch.empty((b, n, 3), dtype=torch.int32, device=dev)) if want_nn else None}
        out.append(lv)
        n = m
    tie_out = torch.empty((b,), dtype=torch.int32, device=dev)
    ia = lambda vals: (ctypes.c_int * nlev)(*[int(v) for v in vals])  # noqa: E731
    pa = lambda ts: (ctypes.c_void_p * nlev)(*[None if t is None else t.data_ptr() for t in ts])  # noqa: E731
    with torch.cuda.device(dev):
        check(lib.pn2_coarse_geometry(b, n0, nlev, ia(npoints), (ctypes.c_float * nlev)(*[float(r) for r in radii]), ia(nsamples),
                                      ptr(xyz0), ptr(tie_in), pa([lv["fps_idx"] for lv in out]), pa([lv["new_xyz"] for lv in out]),
                                      pa([lv["idx"] for lv in out]), pa([lv["cnt"] for lv in out]),
                                      pa([lv["nn"][0] if want_nn else None for lv in out]),
                                      pa([lv["nn"][1] if want_nn else None for lv in out]), ptr(tie_out),
                                      config.fps_mode(fps_arith_mode), config.bq_mode(bq_arith_mode), stream_ptr()),
              "pn2_coarse_geometry")
    tag_fps_output(out[-1]["new_xyz"], tie_out, fps_arith_mode)  # a further level may nest on the last one
    return out


def sa_features_inference(xyz, new_xyz, points, idx, mlp, bn=True, bn_decay=None):
    """The feature half of an SA layer (inference, max pooling): gather + MLP + max over K,
    fused when the widths allow it.  Must be called inside the layer's variable scope.
    -> (B,npoint,mlp[-1])"""
    nsample = idx.shape[2]
    new_points = None
    if USE_FUSED_SA:
        new_points = _sa_fused_inference(xyz, new_xyz, points, idx, mlp, bn, "conv%d")
    if new_points is None:  # every other kernel reads dense rows (the fused kernel above takes column blocks in place)
        xyz = xyz.contiguous()
        points = None if points is None else points.contiguous()
    if new_points is None and points is not None and points.dtype == torch.bfloat16:
        points = points.float()  # configuration outside the bf16 kernel: run the fp32 kernels on the exact values
        if USE_FUSED_SA:
            new_points = _sa_fused_inference(xyz, new_xyz, points, idx, mlp, bn, "conv%d")
    if new_points is None and USE_FUSED_SA and len(mlp) == 3 and mlp[0] == mlp[1] == 128 and mlp[2] % 32 == 0:
        # [128,128,wide]: gather + first two layers fused (activations stay in registers), the wide
        # last layer + max over K on pn2_linear
        h = _sa_fused_inference(xyz, new_xyz, points, idx, mlp[:2], bn, "conv%d", pool=False)
        if h is not None:
            h = tf_util.conv2d(h, mlp[2], [1, 1], padding="VALID", stride=[1, 1], bn=bn, is_training=False,
                               scope="conv2", bn_decay=bn_decay, pool=nsample)
            new_points = h.squeeze(2)
    if (new_points is None and USE_MLP_WIDE and USE_FUSED_SA and nsample == 32 and points is not None
            and points.dtype == torch.float32 and points.shape[2] % 4 == 0 and 1 <= len(mlp) <= 3
            and all(w in (128, 256, 512) for w in mlp) and idx.shape[0] * idx.shape[1] * 32 >= WIDE_MIN_ROWS):
        # all layers wide (SA4): gather + centre + concat + the whole MLP + max over K in one launch
        folded, cprev = [], 3 + points.shape[2]
        for i, cout in enumerate(mlp):
            with tf_util.variable_scope("conv%d" % i):
                if i == 0:  # the kernel's first-layer row order: [features | xyz | zero pad to a multiple of 8]
                    folded.append(tf_util.folded_dense(cprev, cout, bn, (1, 1, cprev, cout), pad_to=32,
                                                       pad_in=-(-cprev // 8) * 8, rotate_rows=cprev - 3))
                else:
                    folded.append(tf_util.folded_dense(cprev, cout, bn, (1, 1, cprev, cout), pad_to=32))
            cprev = cout
        if USE_HOISTED_SA:
            new_points = tf_util.hip_sa_mlp_wide_pre(xyz, new_xyz, points, idx, [f[0] for f in folded], [f[1] for f in folded])
        if new_points is None:
            new_points = tf_util.hip_sa_mlp_wide(xyz, new_xyz, points, idx, [f[0] for f in folded], [f[1] for f in folded])
    if new_points is None:
        h = _sa_group_concat(xyz, new_xyz, points, idx)  # (B,M,K,3+C)
        pool_ok = nsample == 16 or nsample % 32 == 0
        for i, cout in enumerate(mlp):
            last = i == len(mlp) - 1
            h = tf_util.conv2d(h, cout, [1, 1], padding="VALID", stride=[1, 1], bn=bn, is_training=False,
                               scope="conv%d" % i, bn_decay=bn_decay, pool=nsample if (last and pool_ok) else 0)
        if not pool_ok:
            h = h.amax(dim=2, keepdim=True)
        new_points = h.squeeze(2)
    return new_points


_POOL_MODES = {"max": 0, "avg": 1, "weighted_avg": 2, "max_and_avg": 3}


class _GroupPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gxyz, mode):
        from .._lib import check, lib, ptr, stream_ptr
        b, m, k, c = x.shape
        x = x.contiguous()
        gxyz = gxyz.contiguous() if gxyz is not None else None
        out = torch.empty((b, m, 1, 2 * c if mode == 3 else c), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            check(lib.pn2_group_pool(b * m, k, c, mode, ptr(x), ptr(gxyz), ptr(out), stream_ptr()), "pn2_group_pool")
        ctx.save_for_backward(x, gxyz if gxyz is not None else x.new_empty(0))
        ctx.mode = mode
        return out

    @staticmethod
    def backward(ctx, dout):
        from .._lib import check, lib, ptr, stream_ptr
        x, gxyz = ctx.saved_tensors
        b, m, k, c = x.shape
        dx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            check(lib.pn2_group_pool_grad(b * m, k, c, ctx.mode, ptr(x), ptr(gxyz if gxyz.numel() else None),
                                          ptr(dout.contiguous()), ptr(dx), stream_ptr()), "pn2_group_pool_grad")
        return dx, None, None


def group_pool(new_points, grouped_xyz, pooling):
    """(B,M,K,C) -> (B,M,1,C) for max / avg / weighted_avg, (B,M,1,2C) = [avg | max] for max_and_avg, on pn2_group_pool
    (pointnet_util.py:165-191 of the reference); differentiable w.r.t. new_points."""
    mode = _POOL_MODES[pooling]
    require_cuda(new_points)
    if new_points.dtype != torch.float32:
        raise TypeError("pooling expects float32 features")
    tf_util.assert_not_deferred(new_points, "a pooling")
    return _GroupPool.apply(new_points, grouped_xyz if mode == 2 else None, mode)


def pointnet_sa_module(xyz, points, npoint, radius, nsample, mlp, mlp2, group_all, is_training, bn_decay, scope,
                       bn=True, pooling="max", knn=False, use_xyz=True, use_nchw=False, geometry=None):
    """PointNet Set Abstraction module -> new_xyz (B,npoint,3), new_points (B,npoint,mlp[-1] or mlp2[-1]), idx.
    geometry = (new_xyz, idx[, plan]) (extension, training path): precomputed FPS / ball-query result of this level
    (+ the scatter plan of the grouping's gradient); (new_xyz, None) (inference): only the samples are given."""
    require_cuda(xyz, points)
    # use_nchw: accepted and ignored.  In the reference it only transposes the grouped tensor to NCHW around the conv
    # stack and back (pointnet_util.py:143-146,165-166: a cuDNN layout hint, "faster than NHWC"); values, shapes and variable
    # names are the same either way, and these kernels are channels-last by construction.
    with tf_util.variable_scope(scope):
        if not is_training and not group_all and not knn and use_xyz and pooling == "max":
            # ---- inference fast path: HIP index ops + fused / MFMA MLP -------------
            # (xyz / points may be column blocks of a wider batch -- point_cloud[:, :, 0:3] / [:, :, 3:6]: the sampler, the
            # ball query and the fused MLP kernel read them where they lie, nothing is copied)
            if geometry is not None and geometry[1] is None:
                # the samples of this level were drawn ahead (runtime.StaggeredPipeline: the sampling of a batch is a graph of
                # its own, submitted ahead of the batch's dense half); the ball query is still this module's
                new_xyz = geometry[0]
                if len(geometry) > 2 and geometry[2] is not None:  # the cloud was binned beside the sampling (model.sa1_samples)
                    idx, _ = tf_grouping.query_ball_point_binned(radius, nsample, xyz, new_xyz, geometry[2])
                else:
                    idx, _ = query_ball_point(radius, nsample, xyz, new_xyz)
            else:
                new_xyz, idx = geometry[:2] if geometry is not None else sa_geometry(xyz, npoint, radius, nsample)
            new_points = sa_features_inference(xyz, new_xyz, points, idx, mlp, bn, bn_decay)
            new_points = new_points.unsqueeze(2)
        else:
            if geometry is not None and geometry[1] is None:
                raise ValueError("geometry=(new_xyz, None) (samples drawn ahead, ball query here) is the inference fast path's "
                                 "hand-over: pass the idx too for training / group_all / knn / non-max pooling")
            # training, features of >= 16 channels, geometry + scatter plan computed ahead: the feature half of the first
            # conv runs on the n SOURCE points instead of the m * nsample grouped rows (tf_util._TrainHoistedBnRelu)
            hoist = (bool(is_training) and tf_util.USE_HOISTED_TRAIN and not group_all and not knn and use_xyz and bn
                     and pooling == "max" and points is not None and points.dtype == torch.float32 and points.shape[2] >= 16
                     and len(mlp) > 0 and mlp[0] % 4 == 0 and nsample <= 1024 and geometry is not None and len(geometry) > 2
                     and _plan_usable(geometry[2], mlp[0]))
            first = 0
            # a layer whose output feeds only the next conv2d of this loop may hand over its UN-normalised output
            # (tf_util.USE_BN_ON_LOAD): the next layer applies the batch norm + ReLU while loading it
            defer = lambda i, rows: (bool(is_training) and bn and i + 1 < len(mlp)  # noqa: E731
                                     and tf_util.can_defer_bn(rows, mlp[i], mlp[i + 1]))
            small_first = (bool(is_training) and tf_util.USE_SA_FIRST_LAYER_FUSED and tf_util.USE_BN_FINISH_IN_PRODUCER
                           and not hoist and not group_all and not knn and use_xyz and bn and pooling == "max"
                           and points is not None and points.dtype == torch.float32 and points.shape[2] <= 5
                           and not points.requires_grad and len(mlp) > 0 and mlp[0] % 4 == 0 and mlp[0] <= 1024
                           and nsample <= 1024)
            if small_first:
                # few point channels (the level-0 module: xyz + rgb): front end + first conv + its statistics in ONE launch
                new_xyz, idx = geometry[:2] if geometry is not None else sa_geometry(xyz, npoint, radius, nsample)
                grouped_xyz = None
                new_points = tf_util.conv2d_sa_first_small(xyz, new_xyz, points, idx, mlp[0], "conv0", bn_decay,
                                                           pool=nsample if len(mlp) == 1 else 0, defer_bn=defer(0, idx.numel()))
                first = 1
            elif hoist:
                new_xyz, idx, grouped_xyz = geometry[0], geometry[1], None
                new_points = tf_util.conv2d_hoisted_first("sa", points, (xyz, new_xyz, idx), geometry[2], 3 + points.shape[2],
                                                          mlp[0], "conv0", bn_decay, pool=nsample if len(mlp) == 1 else 0,
                                                          defer_bn=defer(0, idx.numel()))
                first = 1
            elif group_all:
                nsample = xyz.shape[1]
                new_xyz, new_points, idx, grouped_xyz = sample_and_group_all(xyz, points, use_xyz)
            else:
                new_xyz, new_points, idx, grouped_xyz = sample_and_group(npoint, radius, nsample, xyz, points, knn,
                                                                         use_xyz, geometry=geometry)
            # training: the max over K rides in the last layer's batch-norm kernels (tf_util._TrainDenseBnRelu)
            fuse_pool = bool(is_training) and bn and pooling == "max" and len(mlp) > 0 and nsample <= 1024
            for i, cout in enumerate(mlp):
                if i < first:
                    continue
                new_points = tf_util.conv2d(new_points, cout, [1, 1], padding="VALID", stride=[1, 1], bn=bn,
                                            is_training=is_training, scope="conv%d" % i, bn_decay=bn_decay,
                                            pool=nsample if (fuse_pool and i == len(mlp) - 1) else 0,
                                            defer_bn=defer(i, new_points.numel() // new_points.shape[-1]))
            if pooling not in _POOL_MODES:
                raise ValueError("unknown pooling %r" % pooling)
            if not (pooling == "max" and fuse_pool):
                new_points = group_pool(new_points, grouped_xyz, pooling)  # HIP kernel (:165-191), (B,M,K,C) -> (B,M,1,C')
        if mlp2 is not None:
            for i, cout in enumerate(mlp2):
                new_points = tf_util.conv2d(new_points, cout, [1, 1], padding="VALID", stride=[1, 1], bn=bn,
                                            is_training=is_training, scope="conv_post_%d" % i, bn_decay=bn_decay)
        new_points = new_points.squeeze(2)
        return new_xyz, new_points, idx


def pointnet_sa_module_msg(xyz, points, npoint, radius_list, nsample_list, mlp_list, is_training, bn_decay, scope,
                           bn=True, use_xyz=True, use_nchw=False, new_xyz=None):
    """Multi-scale grouping SA module (:219-282): one FPS, per scale ball query +
    group + MLP + max; NOTE the concat order here is [features, xyz] (:259), unlike
    sample_and_group.  -> new_xyz, new_points (B,npoint,sum mlp[k][-1]).
    new_xyz (extension): the npoint samples of this very cloud, drawn ahead (runtime.StaggeredPipeline runs a batch's sampling as
    a graph of its own); everything else is done here."""
    require_cuda(xyz, points)
    with tf_util.variable_scope(scope):
        xyz = xyz.contiguous()
        if new_xyz is None:
            fps_idx, fps_tie, _ = farthest_point_sample_with_ties(npoint, xyz)  # (:36-37; the tie record rides on new_xyz, see
            new_xyz = tag_fps_output(gather_point(xyz, fps_idx), fps_tie)         #  tf_sampling.USE_NESTED_FPS)
        elif tuple(new_xyz.shape) != (xyz.shape[0], int(npoint), 3):
            raise ValueError("new_xyz: (batch_size, npoint, 3) samples of xyz expected")
        outs = []
        # one scan of xyz for all radii (the reference re-scans once per radius, :245-250)
        queries = query_ball_point_multi(radius_list, nsample_list, xyz, new_xyz)
        for i, (radius, nsample) in enumerate(zip(radius_list, nsample_list)):
            idx = queries[i][0]
            if (not is_training) and USE_FUSED_SA and (use_xyz or points is None):
                # gather + MLP + max in one kernel; this module's [features | xyz] order is absorbed into the weights
                fused = _sa_fused_inference(xyz, new_xyz, points, idx, mlp_list[i], bn, "conv%d_" % i + "%d",
                                            xyz_last=True)
                if fused is not None:
                    outs.append(fused)
                    continue
            grouped_xyz = group_point(xyz, idx) - new_xyz.unsqueeze(2)
            if points is not None:
                grouped_points = group_point(points, idx)
                if use_xyz:
                    grouped_points = torch.cat([grouped_points, grouped_xyz], dim=-1)
            else:
                grouped_points = grouped_xyz
            pool_ok = (not is_training) and (nsample == 16 or nsample % 32 == 0)
            for j, cout in enumerate(mlp_list[i]):
                last = j == len(mlp_list[i]) - 1
                grouped_points = tf_util.conv2d(grouped_points, cout, [1, 1], padding="VALID", stride=[1, 1], bn=bn,
                                                is_training=is_training, scope="conv%d_%d" % (i, j),
                                                bn_decay=bn_decay, pool=nsample if (last and pool_ok) else 0)
            if pool_ok:
                outs.append(grouped_points.squeeze(2))
            else:
                outs.append(grouped_points.amax(dim=2))
        return new_xyz, torch.cat(outs, dim=-1)


def _fp_interp_concat(dist, idx, points1, points2, pad_to=1):
    """-> (B, n, roundup(c2+c1, pad_to)); pad columns are zero."""
    b, n, _ = dist.shape
    m, c2 = points2.shape[1], points2.shape[2]
    c1 = 0 if points1 is None else points1.shape[2]
    cw = -(-(c1 + c2) // pad_to) * pad_to
    out = torch.empty((b, n, cw), dtype=torch.float32, device=dist.device)
    p1 = None if points1 is None else points1.contiguous()
    p2 = points2.contiguous()
    with torch.cuda.device(dist.device):
        check(lib.pn2_fp_interp_concat(b, n, m, c1, c2, ptr(dist), ptr(idx), ptr(p1), ptr(p2), ptr(out), cw,
                                       stream_ptr()), "pn2_fp_interp_concat")
    return out


USE_FUSED_TRAIN_FRONT = True  # set False: the training path's SA / FP front ends as separate ops (tests / A-B)


USE_MLP_CHAIN = True  # set False to force one pn2_linear launch per layer (tests/bench)


USE_MLP_WIDE = True   # set False: coarse-level MLPs as one pn2_linear launch per layer (tests / A-B)
WIDE_MIN_ROWS = 4096  # FP2 (128 tiles) still wins with its front end fused in (28 us vs 8 + 25); FP1 (1024 rows = 32 tiles) does not


USE_FUSED_FP = True    # set False to force pn2_fp_interp_concat + separate MLP launches (tests/bench)


def dense_mlp_inference(x2d, cin, mlp, scope_fmt, bn=True, fp_front=None):
    """Run a stack of 1x1-conv layers on (rows, cin_padded) rows.  Consecutive layers whose widths are
    <= 128 run as LDS-resident chains (pn2_mlp_chain, up to 2 layers per launch); anything else runs
    one pn2_linear per layer.  `cin` is the true input width (x2d may carry zero pad columns).
    fp_front = (dist, idx, points1, points2, pad_to): the rows are the FP front end; when the first layers
    qualify they are produced inside the first chain kernel (pn2_fp_mlp_fused) and x2d may be None,
    otherwise they are materialised with pn2_fp_interp_concat.
    Must be called inside the module's variable scope."""
    if x2d is None:
        dist, idx, points1, points2, pad_to = fp_front
        rows = dist.shape[0] * dist.shape[1]
        cw = -(-cin // pad_to) * pad_to
    else:
        rows, cw = x2d.shape
    folded = []
    c = cin
    for i, cout in enumerate(mlp):
        with tf_util.variable_scope(scope_fmt % i):
            pad_in = cw if i == 0 else None
            folded.append(tf_util.folded_dense(c, cout, bn, (1, 1, c, cout), pad_to=32, pad_in=pad_in))
        c = cout
    i, h = 0, x2d
    if h is None:
        if (USE_FUSED_FP and USE_MLP_CHAIN and USE_HOISTED_FP and rows >= 65536 and len(mlp) >= 2
                and all(w <= 128 and w % 32 == 0 for w in mlp[:3])):
            # first layer hoisted by linearity (interp(points2) @ W == interp(points2 @ W)): points2 @ W1a on the m known
            # rows, then front end + up to three LDS-resident layers in one kernel
            take = min(3, len(mlp))
            y = tf_util.hip_fp_mlp_fused_pre(dist, idx, points1, points2, [folded[k][0] for k in range(take)],
                                             [folded[k][1] for k in range(take)])
            if y is not None:
                h, i = y, take
        if h is None and points1 is not None:
            points1 = points1.contiguous()  # (only the hoisted chain above reads a column block in place)
        if (h is None and USE_FUSED_FP and USE_MLP_CHAIN and rows >= 65536 and points2.shape[2] % 8 == 0
                and mlp[0] <= 128 and mlp[0] % 32 == 0):
            for take in (2, 1):
                if take > len(mlp) or any(w > 128 or w % 32 for w in mlp[:take]):
                    continue
                y = tf_util.hip_fp_mlp_fused(dist, idx, points1, points2, [folded[k][0] for k in range(take)],
                                             [folded[k][1] for k in range(take)])
                if y is not None:
                    h, i = y, take
                    break
        if (h is None and USE_FUSED_FP and USE_MLP_WIDE and WIDE_MIN_ROWS <= rows <= 65536 and dist.shape[1] % 32 == 0
                and points2.shape[2] % 4 == 0 and (points1 is None or points1.shape[2] % 4 == 0)
                and all(w in (128, 256, 512) for w in mlp[:3])):
            # coarse FP levels: front end + the whole MLP in one launch (pn2_fp_mlp_wide)
            take = min(3, len(mlp))
            y = None
            if USE_HOISTED_FP and points1 is not None:
                y = tf_util.hip_fp_mlp_wide_pre(dist, idx, points1, points2, [folded[k][0] for k in range(take)],
                                                [folded[k][1] for k in range(take)])
            if y is None:
                y = tf_util.hip_fp_mlp_wide(dist, idx, points1, points2, [folded[k][0] for k in range(take)],
                                            [folded[k][1] for k in range(take)])
            if y is not None:
                h, i = y, take
        if h is None:
            x = _fp_interp_concat(dist, idx, points1, points2, pad_to=pad_to)
            h = x.reshape(rows, x.shape[2])
    while i < len(mlp):
        done = False
        # coarse levels (4096 .. 32768 rows, widths 128 / 256 / 512): the remaining layers in one launch (pn2_mlp_wide)
        if USE_MLP_WIDE and WIDE_MIN_ROWS <= h.shape[0] <= 65536 and all(w in (128, 256, 512) for w in mlp[i:i + 3]):
            take = min(3, len(mlp) - i)
            y = tf_util.hip_mlp_wide(h, [folded[k][0] for k in range(i, i + take)], [folded[k][1] for k in range(i, i + take)])
            if y is not None:
                h, i = y, i + take
                continue
        # the LDS-resident chain pays off when there are enough 32-row tiles to fill the chip
        if USE_MLP_CHAIN and h.shape[0] >= 65536 and all(w <= 128 and w % 32 == 0 for w in mlp[i:i + 2]):
            for take in (2, 1):
                if i + take > len(mlp) or any(w > 128 or w % 32 for w in mlp[i:i + take]):
                    continue
                y = tf_util.hip_mlp_chain(h, [folded[k][0] for k in range(i, i + take)],
                                          [folded[k][1] for k in range(i, i + take)])
                if y is not None:
                    h, i, done = y, i + take, True
                    break
        if not done:
            w2, b2 = folded[i]
            h = tf_util.hip_linear(h, w2, b2, relu=True)
            if h.shape[1] != mlp[i]:
                h = h[:, :mlp[i]].contiguous()
            i += 1
    return h


def fp_features_inference(dist, idx, points1, points2, mlp, bn=True, bn_decay=None):
    """The feature half of an FP layer (inference) given three_nn's (dist, idx).  Must be called
    inside the layer's variable scope."""
    b, n = dist.shape[0], dist.shape[1]
    cin = points2.shape[2] + (0 if points1 is None else points1.shape[2])
    # rows padded e.g. 131 -> 136 when materialised: 16-byte loads downstream
    h = dense_mlp_inference(None, cin, mlp, "conv_%d", bn, fp_front=(dist, idx, points1, points2, 8))
    return h.reshape(b, n, mlp[-1])


def pointnet_fp_module(xyz1, xyz2, points1, points2, mlp, is_training, bn_decay, scope, bn=True, nn=None, defer_last_bn=0):
    """Feature propagation: xyz1 (B,n1,3) dense, xyz2 (B,n2,3) sparse, points1 (B,n1,c1) or None,
    points2 (B,n2,c2) -> (B,n1,mlp[-1]).  nn = (dist, idx[, plan]) (extension): three_nn(xyz1, xyz2) computed ahead, and
    optionally scatter_plan(idx, n2, dist, weight_kind=2) for the interpolation's gradient.
    defer_last_bn = w > 0 (extension, training): the caller feeds the result to exactly ONE conv1d / conv2d of width w with
    batch norm (model.get_model: the head's fc1) and nothing else -- the last layer then hands over its UN-normalised output
    (tf_util.USE_BN_ON_LOAD) and that layer applies the batch norm + ReLU while loading it."""
    require_cuda(xyz1, xyz2, points1, points2)
    with tf_util.variable_scope(scope):
        dist, idx = (nn[0], nn[1]) if nn is not None else three_nn(xyz1, xyz2)
        plan = nn[2] if nn is not None and len(nn) > 2 else None
        if not is_training:
            # weights + interpolate + concat fused, then LDS-resident MLP chains / MFMA layers
            return fp_features_inference(dist, idx, points1, points2, mlp, bn, bn_decay)
        # the interpolated half of the first conv runs on the n2 known points instead of the n1 dense ones
        # (tf_util._TrainHoistedBnRelu); points1 = a few input channels (the level-0 module) or SA features
        c1 = 0 if points1 is None else points1.shape[2]
        if (tf_util.USE_HOISTED_TRAIN and bn and points1 is not None
                and 1 <= c1 <= 8 and not points1.requires_grad
                and points1.dtype == torch.float32 and points2.dtype == torch.float32 and len(mlp) > 0 and mlp[0] % 32 == 0
                and _plan_usable(plan, mlp[0])):
            rows = points1.shape[0] * points1.shape[1]
            nxt = list(mlp[1:]) + [int(defer_last_bn)]  # width of the layer that consumes layer i (0: unknown -> materialise)
            defer = lambda i: bn and nxt[i] > 0 and tf_util.can_defer_bn(rows, mlp[i], nxt[i])  # noqa: E731
            h = tf_util.conv2d_hoisted_first("fp", points2, (dist, idx, points1), plan, points2.shape[2] + points1.shape[2],
                                             mlp[0], "conv_0", bn_decay, defer_bn=defer(0))
            for i, cout in enumerate(mlp):
                if i > 0:
                    h = tf_util.conv2d(h, cout, [1, 1], padding="VALID", stride=[1, 1], bn=bn, is_training=is_training,
                                       scope="conv_%d" % i, bn_decay=bn_decay, defer_bn=defer(i))
            return h.squeeze(2)
        if USE_FUSED_TRAIN_FRONT and points2.dtype == torch.float32:
            new_points1 = _FPInterpConcat.apply(dist, idx, None if points1 is None else points1.contiguous(),
                                                points2.contiguous(), plan)
        else:
            dist = torch.clamp(dist, min=1e-10)
            norm = (1.0 / dist).sum(dim=2, keepdim=True)
            weight = (1.0 / dist) / norm
            interpolated = three_interpolate(points2, idx, weight)
            new_points1 = torch.cat([interpolated, points1], dim=2) if points1 is not None else interpolated
        new_points1 = new_points1.unsqueeze(2)
        rows = new_points1.shape[0] * new_points1.shape[1]
        nxt = list(mlp[1:]) + [int(defer_last_bn)]
        for i, cout in enumerate(mlp):
            # un-normalised hand-over to the next layer of the stack (tf_util.USE_BN_ON_LOAD)
            dfr = bool(is_training) and bn and nxt[i] > 0 and tf_util.can_defer_bn(rows, cout, nxt[i])
            new_points1 = tf_util.conv2d(new_points1, cout, [1, 1], padding="VALID", stride=[1, 1], bn=bn,
                                         is_training=is_training, scope="conv_%d" % i, bn_decay=bn_decay, defer_bn=dfr)
        return new_points1.squeeze(2)
