"""query_ball_point / group_point on the MI355X.

Same names, argument order and shapes as the reference wrappers
(tf_ops/tf_grouping.py:13-25 query_ball_point, :46-54 group_point, gradient :57-61).
knn_point / select_top_k (tf_grouping.py:31-43,64-89) are not on the model's path
(model always passes knn=False) and are listed as "next" in DESIGN.md.
"""
import torch

from .. import config
from .._lib import PN2_EUNSUP, check, lib, ptr, require_cuda, rows_in_place, stream_ptr


def query_ball_point(radius, nsample, xyz1, xyz2, kernel=0, arith_mode=None):
    """radius float, nsample int, xyz1 (b,n,3) dataset, xyz2 (b,m,3) queries
    -> idx (b,m,nsample) int32, pts_cnt (b,m) int32.  Not differentiable.
    kernel (extension, tests / diagnostics): 0 = chosen by shape, 1 / 2 / 3 = a specific kernel
    (pn2_query_ball_point_kernel); every kernel returns the same bits.
    arith_mode (extension): contraction of the squared-distance expression, config.BQ_ARITH_DEFAULT when None."""
    if not radius > 0:
        raise ValueError("QueryBallPoint expects positive radius")  # tf_grouping.cpp:80-83
    if nsample <= 0:
        raise ValueError("QueryBallPoint expects positive nsample")  # tf_grouping.cpp:84-87
    require_cuda(xyz1, xyz2)
    for t, nm in ((xyz1, "xyz1"), (xyz2, "xyz2")):
        if t.dim() != 3 or t.shape[2] != 3:
            raise ValueError("QueryBallPoint expects (batch_size, ndataset, 3) %s shape." % nm)  # :92-106
        if t.dtype != torch.float32:
            raise TypeError("QueryBallPoint expects float32 %s" % nm)
    if xyz1.shape[0] != xyz2.shape[0]:
        raise ValueError("QueryBallPoint expects xyz1 and xyz2 with the same batch size")
    xyz1, ld1 = rows_in_place(xyz1)  # a column block of a wider batch (point_cloud[:, :, 0:3]) is read where it lies
    xyz2 = xyz2.detach().contiguous()
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=xyz1.device)
    cnt = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        if ld1 != 3:
            rc = PN2_EUNSUP if kernel else lib.pn2_query_ball_point_ld(
                b, n, m, float(radius), int(nsample), ptr(xyz1), ld1, ptr(xyz2), ptr(idx), ptr(cnt),
                config.bq_mode(arith_mode), stream_ptr())
            if rc != PN2_EUNSUP:
                check(rc, "pn2_query_ball_point_ld")
                return idx, cnt
            xyz1 = xyz1.contiguous()  # no strided kernel for this shape: query a dense copy
        if kernel:
            check(lib.pn2_query_ball_point_kernel(b, n, m, float(radius), int(nsample), ptr(xyz1), ptr(xyz2), ptr(idx),
                                                  ptr(cnt), config.bq_mode(arith_mode), int(kernel), stream_ptr()),
                  "pn2_query_ball_point_kernel")
        else:
            check(lib.pn2_query_ball_point(b, n, m, float(radius), int(nsample), ptr(xyz1), ptr(xyz2), ptr(idx),
                                           ptr(cnt), config.bq_mode(arith_mode), stream_ptr()), "pn2_query_ball_point")
    return idx, cnt


BIN_MAX_N, BIN_MIN_N, BIN_MIN_M, BIN_MAX_NSAMPLE = 8192, 4096, 256, 64  # the LDS-grid kernel's range (pn2_grouping.hip)


def _version_of(t):
    """the tensor's version counter, or None for tensors that keep none (created under torch.inference_mode())"""
    try:
        return t._version
    except RuntimeError:
        return None


def ball_query_bin_alloc(xyz1):
    """workspace of ball_query_bin for xyz1 (b,n,3), or None when n is outside the grid kernel's range"""
    b, n, _ = xyz1.shape
    if not (BIN_MIN_N <= n <= BIN_MAX_N):
        return None
    stride = int(lib.pn2_ball_query_bin_bytes(n))
    ws = torch.empty((b * stride + 256,), dtype=torch.uint8, device=xyz1.device)
    off = (-ws.data_ptr()) % 256
    return ws[off:off + b * stride]


def ball_query_bin(radius, xyz1, out=None):
    """Extension: sort every cloud of xyz1 (b,n,3) into the uniform grid of `radius` ONCE (pn2_ball_query_bin, one workgroup
    per cloud) -> an opaque uint8 tensor for query_ball_point_binned, or None when the shape is outside the grid kernel's
    range (the caller then uses query_ball_point).  Depends on xyz1 only, so it can run beside the FPS of the same level."""
    if not radius > 0:
        raise ValueError("QueryBallPoint expects positive radius")
    require_cuda(xyz1)
    if xyz1.dim() != 3 or xyz1.shape[2] != 3 or xyz1.dtype != torch.float32:
        raise ValueError("QueryBallPoint expects (batch_size, ndataset, 3) float32 xyz1")
    b, n, _ = xyz1.shape
    if not (BIN_MIN_N <= n <= BIN_MAX_N):
        return None
    src = (xyz1.data_ptr(), _version_of(xyz1))  # the caller's tensor
    # a column block of a wider batch (point_cloud[:, :, 0:3]) is read where it lies; other layouts are copied dense
    xyz1, ld1 = rows_in_place(xyz1)
    bins = out if out is not None else ball_query_bin_alloc(xyz1)  # `out`: allocated by the caller (on ITS stream)
    with torch.cuda.device(xyz1.device):
        check(lib.pn2_ball_query_bin_ld(b, n, float(radius), ptr(xyz1), ld1, ptr(bins), bins.numel(), stream_ptr()),
              "pn2_ball_query_bin_ld")
    # what the bins describe: the C entry point takes an opaque pointer and cannot check it -- bins of a smaller
    # cloud would be read out of bounds, bins of a smaller radius would silently miss neighbours
    # ... and bins of ANOTHER cloud of the same shape (a reused `out=` workspace after the input batch changed) would return wrong
    # neighbours: the tag also names the storage and the version of the cloud it was built from.  Writes through
    # raw pointers -- this library's own kernels, a graph replay into a static buffer -- do not bump a tensor's version: whoever
    # overwrites xyz1 that way must bin again.
    bins._pn2_bins_of = (float(radius), b, n) + src
    return bins


def query_ball_point_binned(radius, nsample, xyz1, xyz2, bins, arith_mode=None):
    """query_ball_point(radius, nsample, xyz1, xyz2) on clouds binned by ball_query_bin(radius, xyz1): bit-identical result,
    the query workgroups copy the cell-sorted cloud instead of each re-binning it."""
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    if bins is None or nsample > BIN_MAX_NSAMPLE or m < BIN_MIN_M:
        return query_ball_point(radius, nsample, xyz1, xyz2, arith_mode=arith_mode)
    require_cuda(xyz1, xyz2)
    src = (xyz1.data_ptr(), _version_of(xyz1))
    xyz1 = xyz1.detach()   # (never read by the kernel: the bins hold the cell-sorted cloud; a strided view is fine)
    xyz2 = xyz2.detach().contiguous()
    made = getattr(bins, "_pn2_bins_of", None)
    if made is None:
        raise ValueError("bins must come from ball_query_bin(radius, xyz1)")
    if made[:3] != (float(radius), b, n):
        raise ValueError("bins were built for radius %g on a (%d, %d, 3) cloud: they do not describe this query (radius %g, "
                         "cloud (%d, %d, 3))" % (made[0], made[1], made[2], radius, b, n))
    if made[3:] != src:
        raise ValueError("bins were built from another cloud (or xyz1 was modified in place since): bin again")
    if bins.dtype != torch.uint8 or bins.data_ptr() % 256 != 0 or bins.numel() < b * int(lib.pn2_ball_query_bin_bytes(n)):
        raise ValueError("bins: expected a 256-byte aligned uint8 tensor of b * pn2_ball_query_bin_bytes(n) bytes")
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=xyz1.device)
    cnt = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        check(lib.pn2_query_ball_point_binned(b, n, m, float(radius), int(nsample), ptr(xyz1), ptr(xyz2), ptr(bins), ptr(idx),
                                              ptr(cnt), config.bq_mode(arith_mode), stream_ptr()), "pn2_query_ball_point_binned")
    return idx, cnt


def query_ball_point_multi(radius_list, nsample_list, xyz1, xyz2, arith_mode=None):
    """[(idx, pts_cnt) for each (radius, nsample)] from ONE scan of xyz1 (pn2_query_ball_point_multi): what
    pointnet_sa_module_msg needs (util/pointnet_util.py:245-250 runs query_ball_point once per radius).
    Bit-identical to separate query_ball_point calls; falls back to them when the configuration is unsupported."""
    import ctypes
    from .._lib import PN2_EUNSUP
    if len(radius_list) != len(nsample_list) or len(radius_list) == 0:
        raise ValueError("radius_list and nsample_list must have the same non-zero length")
    for radius, nsample in zip(radius_list, nsample_list):
        if not radius > 0:
            raise ValueError("QueryBallPoint expects positive radius")
        if nsample <= 0:
            raise ValueError("QueryBallPoint expects positive nsample")
    require_cuda(xyz1, xyz2)
    R = len(radius_list)
    if R > 1 and xyz1.dim() == 3 and xyz2.dim() == 3 and xyz1.dtype == torch.float32 and xyz2.dtype == torch.float32:
        x1, x2 = xyz1.detach().contiguous(), xyz2.detach().contiguous()
        b, n, _ = x1.shape
        m = x2.shape[1]
        idxs = [torch.empty((b, m, int(k)), dtype=torch.int32, device=x1.device) for k in nsample_list]
        cnts = [torch.empty((b, m), dtype=torch.int32, device=x1.device) for _ in nsample_list]
        radii = (ctypes.c_float * R)(*[float(r) for r in radius_list])
        nss = (ctypes.c_int * R)(*[int(k) for k in nsample_list])
        ip = (ctypes.c_void_p * R)(*[t.data_ptr() for t in idxs])
        cp = (ctypes.c_void_p * R)(*[t.data_ptr() for t in cnts])
        with torch.cuda.device(x1.device):
            rc = lib.pn2_query_ball_point_multi(b, n, m, R, ctypes.cast(radii, ctypes.c_void_p), ctypes.cast(nss, ctypes.c_void_p),
                                                ptr(x1), ptr(x2), ctypes.cast(ip, ctypes.c_void_p),
                                                ctypes.cast(cp, ctypes.c_void_p), config.bq_mode(arith_mode), stream_ptr())
        if rc != PN2_EUNSUP:
            check(rc, "pn2_query_ball_point_multi")
            return list(zip(idxs, cnts))
    return [query_ball_point(r, k, xyz1, xyz2, arith_mode=arith_mode) for r, k in zip(radius_list, nsample_list)]


class _GroupPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx):
        b, n, c = points.shape
        _, m, ns = idx.shape
        out = torch.empty((b, m, ns, c), dtype=torch.float32, device=points.device)
        with torch.cuda.device(points.device):
            check(lib.pn2_group_point(b, n, c, m, ns, ptr(points), ptr(idx), ptr(out), stream_ptr()),
                  "pn2_group_point")
        ctx.save_for_backward(idx)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        b, m, ns, c = grad_out.shape
        gp = torch.empty((b, ctx.n, c), dtype=torch.float32, device=grad_out.device)
        nbytes = lib.pn2_group_point_grad_workspace_bytes(b, ctx.n, m, ns)
        ws = torch.empty(nbytes // 4, dtype=torch.int32, device=grad_out.device)
        with torch.cuda.device(grad_out.device):
            check(lib.pn2_group_point_grad_ws(b, ctx.n, c, m, ns, ptr(grad_out), ptr(idx), ptr(gp), ptr(ws), nbytes,
                                              stream_ptr()), "pn2_group_point_grad_ws")
        return gp, None


def group_point(points, idx):
    """points (b,n,c) float32, idx (b,m,nsample) int32 -> (b,m,nsample,c); gradient w.r.t. points."""
    require_cuda(points, idx)
    if points.dim() != 3:
        raise ValueError("GroupPoint expects (batch_size, num_points, channel) points shape")  # tf_grouping.cpp:189-192
    if idx.dim() != 3 or idx.shape[0] != points.shape[0]:
        raise ValueError("GroupPoint expects (batch_size, npoints, nsample) idx shape")  # :199-203
    if points.dtype != torch.float32 or idx.dtype != torch.int32:
        raise TypeError("GroupPoint expects float32 points and int32 idx")
    return _GroupPoint.apply(points.contiguous(), idx.contiguous())


def select_top_k(k, dist):
    """k int, dist (b,m,n) float32 -> idx (b,m,n) int32, dist_out (b,m,n): the first k of every row are
    the k smallest in ascending order (tf_grouping.py:31-40; SelectionSort, not differentiable)."""
    if k <= 0:
        raise ValueError("SelectionSort expects positive k")  # tf_grouping.cpp:142-144
    require_cuda(dist)
    if dist.dim() != 3:
        raise ValueError("SelectionSort expects (b,m,n) dist shape.")  # tf_grouping.cpp:152-154
    if dist.dtype != torch.float32:
        raise TypeError("SelectionSort expects float32 dist")
    dist = dist.detach().contiguous()
    b, m, n = dist.shape
    outi = torch.empty((b, m, n), dtype=torch.int32, device=dist.device)
    out = torch.empty((b, m, n), dtype=torch.float32, device=dist.device)
    with torch.cuda.device(dist.device):
        check(lib.pn2_selection_sort(b, n, m, int(k), ptr(dist), ptr(outi), ptr(out), stream_ptr()),
              "pn2_selection_sort")
    return outi, out


def knn_point(k, xyz1, xyz2):
    """k int, xyz1 (b,n,c) dataset, xyz2 (b,m,c) queries -> val (b,m,k) squared L2, idx (b,m,k) int32
    (tf_grouping.py:64-89: tile + subtract + square + reduce_sum, then select_top_k).  The distance
    matrix is built with separate fp32 ops summed left to right, as the separate TF ops do."""
    require_cuda(xyz1, xyz2)
    d = (xyz1.detach().unsqueeze(1) - xyz2.detach().unsqueeze(2)) ** 2  # (b,m,n,c)
    dist = d[..., 0]
    for ch in range(1, d.shape[-1]):
        dist = dist + d[..., ch]
    outi, out = select_top_k(k, dist)
    return out[:, :, :k].contiguous(), outi[:, :, :k].contiguous()
