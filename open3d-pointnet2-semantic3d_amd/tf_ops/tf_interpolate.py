"""three_nn / three_interpolate on the MI355X.

Same names, argument order and shapes as the reference wrappers
(tf_ops/tf_interpolate.py:13-22 three_nn, :50-59 three_interpolate, gradient
:62-72).  The reference runs these on the CPU (tf_interpolate.cpp:184,283,378,482
DEVICE_CPU); here they stay on the GPU.
"""
import torch

from .._lib import check, lib, ptr, require_cuda, stream_ptr


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
    xyz1 = xyz1.detach().contiguous()
    xyz2 = xyz2.detach().contiguous()
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = torch.empty((b, n, 3), dtype=torch.float32, device=xyz1.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        check(lib.pn2_three_nn(b, n, m, ptr(xyz1), ptr(xyz2), ptr(dist), ptr(idx), stream_ptr()), "pn2_three_nn")
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
        grad_out = grad_out.contiguous()
        b, n, c = grad_out.shape
        gp = torch.empty((b, ctx.m, c), dtype=torch.float32, device=grad_out.device)
        with torch.cuda.device(grad_out.device):
            check(lib.pn2_three_interpolate_grad(b, n, c, ctx.m, ptr(grad_out), ptr(idx), ptr(weight), ptr(gp),
                                                 stream_ptr()), "pn2_three_interpolate_grad")
        return gp, None, None  # idx, weight get no gradient (tf_interpolate.py:66-72)


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
    raise NotImplementedError("interpolate_label_with_color (tf_interpolate.py:28-44) is post-processing outside "
                              "the SA/FP hot path; see DESIGN.md 'Next'")
