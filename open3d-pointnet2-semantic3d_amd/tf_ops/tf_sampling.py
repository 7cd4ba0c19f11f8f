"""farthest_point_sample / gather_point on the MI355X.

Same names, argument order and shapes as the reference wrappers
(tf_ops/tf_sampling.py:38-46 gather_point, :61-69 farthest_point_sample, gradient
registration :54-58, NoGradient :72), but on torch CUDA(ROCm) tensors through the
C ABI of libpn2_hip.so instead of tf.load_op_library.
"""
import torch

from .. import config
from .._lib import check, lib, ptr, require_cuda, rows_in_place, stream_ptr


def _chk_xyz(t, name, op):
    if t.dim() != 3 or t.shape[2] != 3:
        raise ValueError("%s expects (batch_size,num_points,3) %s shape" % (op, name))  # tf_sampling.cpp:131-134
    if t.dtype != torch.float32:
        raise TypeError("%s expects float32 %s" % (op, name))


# Nested sampling (pn2_fps_nested): the new_xyz an FPS run returns carries that run's tie record as a Python attribute; a
# later farthest_point_sample ON THAT VERY TENSOR (the SA levels chain exactly so, util/pointnet_util.py:36-37) hands the
# record to the kernel, which answers idx = 0..m-1 for every cloud whose parent run met no tie before step m -- the
# sampled answer bit for bit -- and samples the others.  False = always sample (A/B, tests).
USE_NESTED_FPS = True
_TIE_ATTR = "_pn2_fps_tie"


def tag_fps_output(new_xyz, tie, arith_mode=None):
    """mark new_xyz (b,m,3) as the gathered picks, in pick order, of the FPS run (arithmetic mode `arith_mode`) whose tie
    record is `tie` (b,) int32"""
    if tie is not None:
        try:
            setattr(new_xyz, _TIE_ATTR, (tie, new_xyz._version, config.fps_mode(arith_mode)))
        except (RuntimeError, AttributeError):  # inference-mode tensors keep no version counter: no tag, always sample
            pass
    return new_xyz


def drop_fps_tag(t):
    """forget the FPS tie record attached to `t`: for code that overwrites a tensor through
    its raw pointer -- this library's kernels writing into a caller's buffer (tf_util.multi_copy_) -- which no version counter
    sees.  (A graph replay that rewrites new_xyz rewrites its tie record in the same launch: that tag stays true.)  Ball-query bins
    carry their own provenance tag on the bins tensor (tf_grouping.ball_query_bin), keyed on the source's address and version:
    a raw-pointer overwrite of the cloud does not invalidate it -- whoever overwrites a binned cloud that way bins again."""
    for attr in (_TIE_ATTR,):
        if hasattr(t, attr):
            try:
                delattr(t, attr)
            except AttributeError:
                pass
    return t


def fps_tie_record(inp, arith_mode=None):
    """the tie record of the run that produced `inp`, or None (untagged, modified in place since, other batch / device /
    arithmetic mode: the distances of the two levels are only bit-identical under the same contraction)"""
    if not USE_NESTED_FPS:
        return None
    tag = getattr(inp, _TIE_ATTR, None)
    if tag is None:
        return None
    tie, version, mode = tag
    try:
        if inp._version != version or mode != config.fps_mode(arith_mode):
            return None
    except RuntimeError:
        return None
    if tie.device != inp.device or tie.shape[0] != inp.shape[0] or not inp.is_contiguous():
        return None
    return tie


USE_BUCKET_FPS = True  # clouds beyond 16384 points: Morton buckets + bounding-box skipping (pn2_fps_large); False = streaming kernel
FPS_REG_MAX, FPS_BUCKET_MAX = 16384, 131072


def _fps_large(npoint, inp, want_xyz, arith_mode=None):
    import ctypes
    b, n, _ = inp.shape
    out = torch.empty((b, npoint), dtype=torch.int32, device=inp.device)
    new_xyz = torch.empty((b, npoint, 3), dtype=torch.float32, device=inp.device) if want_xyz else None
    wbytes = int(lib.pn2_fps_large_workspace_bytes(b, n))
    ws = torch.empty((wbytes + 256,), dtype=torch.uint8, device=inp.device)
    off = (-ws.data_ptr()) % 256
    with torch.cuda.device(inp.device):
        check(lib.pn2_fps_large(b, n, int(npoint), ptr(inp), ctypes.c_void_p(ws.data_ptr() + off), wbytes, ptr(out),
                                ptr(new_xyz), config.fps_mode(arith_mode), stream_ptr()), "pn2_fps_large")
    return out, new_xyz


def farthest_point_sample(npoint, inp, arith_mode=None):
    """npoint: int; inp (b,n,3) float32 -> (b,npoint) int32.  Not differentiable.
    arith_mode (extension): contraction of the squared-distance expression, config.FPS_ARITH_DEFAULT when None."""
    return farthest_point_sample_with_ties(npoint, inp, arith_mode)[0]


def farthest_point_sample_with_ties(npoint, inp, arith_mode=None, want_xyz=False):
    """farthest_point_sample that also returns the run's tie record for the level below (extension):
    -> idx (b,npoint) int32, tie (b,) int32 or None (clouds beyond 16384 points keep no record), new_xyz or None."""
    if npoint <= 0:
        raise ValueError("FarthestPointSample expects positive npoint")  # tf_sampling.cpp:121-123
    require_cuda(inp)
    _chk_xyz(inp, "inp", "FarthestPointSample")
    tie_in = fps_tie_record(inp, arith_mode)
    b, n, _ = inp.shape
    if n <= FPS_REG_MAX:
        # the register-resident kernels read the cloud once: a column block of a wider batch (point_cloud[:, :, 0:3]) in place
        inp, ld = rows_in_place(inp)
        out = torch.empty((b, npoint), dtype=torch.int32, device=inp.device)
        new_xyz = torch.empty((b, npoint, 3), dtype=torch.float32, device=inp.device) if want_xyz else None
        tie = torch.empty((b,), dtype=torch.int32, device=inp.device) if USE_NESTED_FPS else None
        with torch.cuda.device(inp.device):
            if ld == 3:
                check(lib.pn2_fps_nested(b, n, int(npoint), ptr(inp), None, ptr(out), ptr(new_xyz), ptr(tie_in), ptr(tie),
                                         config.fps_mode(arith_mode), stream_ptr()), "pn2_fps_nested")
            else:
                check(lib.pn2_fps_nested_ld(b, n, int(npoint), ptr(inp), ld, ptr(out), ptr(new_xyz), ptr(tie_in), ptr(tie),
                                            config.fps_mode(arith_mode), stream_ptr()), "pn2_fps_nested_ld")
        return out, tie, new_xyz
    inp = inp.detach().contiguous()
    if USE_BUCKET_FPS and FPS_REG_MAX < n <= FPS_BUCKET_MAX:
        out, new_xyz = _fps_large(npoint, inp, want_xyz, arith_mode)
        return out, None, new_xyz
    out = torch.empty((b, npoint), dtype=torch.int32, device=inp.device)
    new_xyz = torch.empty((b, npoint, 3), dtype=torch.float32, device=inp.device) if want_xyz else None
    tie = torch.empty((b,), dtype=torch.int32, device=inp.device) if USE_NESTED_FPS else None
    temp = None
    if n > 16384:  # PN2_FPS_MAX_REG_POINTS: the streaming kernel needs the reference's (32,n) scratch
        temp = torch.empty((min(b, 32), n), dtype=torch.float32, device=inp.device)
    with torch.cuda.device(inp.device):
        check(lib.pn2_fps_nested(b, n, int(npoint), ptr(inp), ptr(temp), ptr(out), ptr(new_xyz), ptr(tie_in), ptr(tie),
                                 config.fps_mode(arith_mode), stream_ptr()), "pn2_fps_nested")
    return out, tie, new_xyz


class _GatherPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, idx):
        b, n, _ = inp.shape
        m = idx.shape[1]
        out = torch.empty((b, m, 3), dtype=torch.float32, device=inp.device)
        with torch.cuda.device(inp.device):
            check(lib.pn2_gather_point(b, n, m, ptr(inp), ptr(idx), ptr(out), stream_ptr()), "pn2_gather_point")
        ctx.save_for_backward(idx)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, out_g):
        (idx,) = ctx.saved_tensors
        out_g = out_g.contiguous()
        b, m, _ = out_g.shape
        inp_g = torch.empty((b, ctx.n, 3), dtype=torch.float32, device=out_g.device)
        with torch.cuda.device(out_g.device):
            check(lib.pn2_gather_point_grad(b, ctx.n, m, ptr(out_g), ptr(idx), ptr(inp_g), stream_ptr()),
                  "pn2_gather_point_grad")
        return inp_g, None


def gather_point(inp, idx):
    """inp (b,n,3) float32, idx (b,m) int32 -> (b,m,3) float32; gradient w.r.t. inp."""
    require_cuda(inp, idx)
    _chk_xyz(inp, "inp", "GatherPoint")
    if idx.dim() != 2 or idx.shape[0] != inp.shape[0]:
        raise ValueError("GatherPoint expects (batch_size,num_result) idx shape")  # tf_sampling.cpp:175-178
    if idx.dtype != torch.int32:
        raise TypeError("GatherPoint expects int32 idx")
    return _GatherPoint.apply(inp.contiguous(), idx.contiguous())


def farthest_point_sample_and_gather(npoint, inp, arith_mode=None):
    """farthest_point_sample + gather_point in one launch (inference; no gradient):
    -> idx (b,npoint) int32, new_xyz (b,npoint,3) == gather_point(inp, idx) bit for bit."""
    out, tie, new_xyz = farthest_point_sample_with_ties(npoint, inp, arith_mode, want_xyz=True)
    return out, tag_fps_output(new_xyz, tie, arith_mode)


def prob_sample(inp, inpr):
    """tf_ops/tf_sampling.py:18-26: inp (batch_size, ncategory) float32 weights, inpr (batch_size, npoints) float32
    uniforms -> (batch_size, npoints) int32 categories (ProbSample: running sum + binary search, tf_sampling.cu:7-110)."""
    require_cuda(inp, inpr)
    if inp.dim() != 2:
        raise ValueError("ProbSample expects (batch_size,num_choices) inp shape")  # tf_sampling.cpp:86-89
    if inpr.dim() != 2 or inpr.shape[0] != inp.shape[0]:
        raise ValueError("ProbSample expects (batch_size,num_points) inpr shape")  # :92-96
    if inp.dtype != torch.float32 or inpr.dtype != torch.float32:
        raise TypeError("ProbSample expects float32 inputs")
    b, n = inp.shape
    m = inpr.shape[1]
    inp, inpr = inp.detach().contiguous(), inpr.detach().contiguous()
    temp = torch.empty((b, n), dtype=torch.float32, device=inp.device)
    out = torch.empty((b, m), dtype=torch.int32, device=inp.device)
    with torch.cuda.device(inp.device):
        check(lib.pn2_prob_sample(b, n, m, ptr(inp), ptr(inpr), ptr(temp), ptr(out), stream_ptr()), "pn2_prob_sample")
    return out
