"""three_nn / three_interpolate on the MI355X.

Same names, argument order and shapes as the reference wrappers
(tf_ops/tf_interpolate.py:13-22 three_nn, :50-59 three_interpolate, gradient
:62-72).  The reference runs these on the CPU (tf_interpolate.cpp:184,283,378,482
DEVICE_CPU); here they stay on the GPU.
"""
import ctypes

import torch

from .._lib import check, lib, ptr, require_cuda, rows_in_place, stream_ptr


def three_nn(xyz1, xyz2):
    """xyz1 (b,n,3) unknown, xyz2 (b,m,3) known -> dist (b,n,3) float32 SQUARED L2
    ascending, idx (b,n,3) int32.  Not differentiable."""
    require_cuda(xyz1, xyz2)
    for t, nm in ((xyz1, "xyz1"), (xyz2, "xyz2")):
        if t.dim() != 3 or t.shape[2] != 3:
            raise ValueError("ThreeNN expects (b,n,3) %s shape" % nm)  # tf_interpolate.cpp:252-262
        if t.dtype != torch.float32:
            raise TypeError("ThreeNN expects float32 %s" % nm)
    if xyz2.shape[1] < 3:
        raise ValueError("ThreeNN needs at least 3 known points")
    xyz1, ld1 = rows_in_place(xyz1)  # queries that are a column block of a wider batch (point_cloud[:, :, 0:3]): read in place
    xyz2 = xyz2.detach().contiguous()
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = torch.empty((b, n, 3), dtype=torch.float32, device=xyz1.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        if ld1 == 3:
            check(lib.pn2_three_nn(b, n, m, ptr(xyz1), ptr(xyz2), ptr(dist), ptr(idx), stream_ptr()), "pn2_three_nn")
        else:
            check(lib.pn2_three_nn_ld(b, n, m, ptr(xyz1), ld1, ptr(xyz2), ptr(dist), ptr(idx), stream_ptr()), "pn2_three_nn_ld")
    return dist, idx


class _ThreeInterpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx, weight):
        b, m, c = points.shape
        n = idx.shape[1]
        out = torch.empty((b, n, c), dtype=torch.float32, device=points.device)
        with torch.cuda.device(points.device):
            check(lib.pn2_three_interpolate(b, m, c, n, ptr(points), ptr(idx), ptr(weight), ptr(out),
                                            stream_ptr()), "pn2_three_interpolate")
        ctx.save_for_backward(idx, weight)
        ctx.m = m
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        return three_interpolate_grad(grad_out, idx, weight, ctx.m), None, None  # idx, weight get no gradient (tf_interpolate.py:66-72)


def three_interpolate_grad(grad_out, idx, weight, m):
    """gradient of three_interpolate w.r.t. points: (b,n,c) -> (b,m,c) (tf_interpolate.py:62-72), list-and-gather kernel"""
    grad_out = grad_out.contiguous()
    b, n, c = grad_out.shape
    gp = torch.empty((b, m, c), dtype=torch.float32, device=grad_out.device)
    nbytes = lib.pn2_three_interpolate_grad_workspace_bytes(b, n, m)
    ws = torch.empty(nbytes // 4, dtype=torch.int32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        check(lib.pn2_three_interpolate_grad_ws(b, n, c, m, ptr(grad_out), ptr(idx), ptr(weight), ptr(gp), ptr(ws),
                                                nbytes, stream_ptr()), "pn2_three_interpolate_grad_ws")
    return gp


def three_interpolate(points, idx, weight):
    """points (b,m,c), idx (b,n,3) int32, weight (b,n,3) -> (b,n,c); gradient w.r.t. points only."""
    require_cuda(points, idx, weight)
    if points.dim() != 3:
        raise ValueError("ThreeInterpolate expects (b,m,c) points shape")  # tf_interpolate.cpp:339-342
    if idx.dim() != 3 or idx.shape[2] != 3 or idx.shape[0] != points.shape[0]:
        raise ValueError("ThreeInterpolate expects (b,n,3) idx shape")  # :347-351
    if weight.shape != idx.shape:
        raise ValueError("ThreeInterpolate expects (b,n,3) weight shape")  # :353-357
    if points.dtype != torch.float32 or weight.dtype != torch.float32 or idx.dtype != torch.int32:
        raise TypeError("ThreeInterpolate expects float32 points/weight and int32 idx")
    return _ThreeInterpolate.apply(points.contiguous(), idx.contiguous(), weight.detach().contiguous())


def interpolate_label_with_color(sparse_points, sparse_labels, dense_points, knn):
    """tf_ops/tf_interpolate.py:28-44: sparse_points (ns,3) f32, sparse_labels (ns,) int32, dense_points (nd,3) f32,
    knn int -> (dense_labels (nd,) int32, dense_colors (nd,3) uint8): majority label of the knn nearest
    sparse points (exact float64 kNN on the device instead of the reference's CPU KD-tree)."""
    require_cuda(sparse_points, sparse_labels, dense_points)
    if sparse_points.dim() != 2 or sparse_points.shape[1] != 3:
        raise ValueError("sparse_points must be: (num_sparse_points, 3)")  # tf_interpolate.cpp:127-131
    ns = sparse_points.shape[0]
    if sparse_labels.dim() != 1 or sparse_labels.shape[0] != ns:
        raise ValueError("sparse_labels must be: (num_sparse_points, 3)")  # :138-143 (message as in the reference)
    if dense_points.dim() != 2 or dense_points.shape[1] != 3:
        raise ValueError("dense_points must be: (num_dense_points, 3)")  # :148-153
    if not isinstance(knn, int) or isinstance(knn, bool):
        raise ValueError("knn must be an int scalar")  # :159-160
    if knn <= 0:
        raise ValueError("knn must be positive")
    if sparse_points.dtype != torch.float32 or dense_points.dtype != torch.float32 or sparse_labels.dtype != torch.int32:
        raise TypeError("InterpolateLabelWithColor expects float32 points and int32 labels")
    nd = dense_points.shape[0]
    dev = dense_points.device
    sp, sl, dp = sparse_points.contiguous(), sparse_labels.contiguous(), dense_points.contiguous()
    labels = torch.empty((nd,), dtype=torch.int32, device=dev)
    colors = torch.empty((nd, 3), dtype=torch.uint8, device=dev)
    wbytes = int(lib.pn2_interpolate_label_workspace_bytes(ns))
    ws = torch.empty((wbytes + 256,), dtype=torch.uint8, device=dev)
    off = (-ws.data_ptr()) % 256
    with torch.cuda.device(dev):
        check(lib.pn2_interpolate_label_with_color(ns, nd, ptr(sp), ptr(sl), ptr(dp), ptr(labels), ptr(colors), knn,
                                                   ctypes.c_void_p(ws.data_ptr() + off), wbytes, stream_ptr()),
              "pn2_interpolate_label_with_color")
    return labels, colors
