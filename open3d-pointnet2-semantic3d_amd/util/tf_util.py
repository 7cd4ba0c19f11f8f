"""Layer helpers with the reference's names and maths, torch/HIP underneath.

Mirrors the subset of util/tf_util.py the SA/FP stack uses: conv2d (1x1 only,
:128-204), conv1d (kernel 1 only, :54-125), batch norm (:555-581, epsilon 1e-3,
center+scale, moving averages updated in place), dropout (:646-665), xavier
weights / zero biases (:30-51,109-111).  TF's implicit variable scopes become an
explicit VariableStore (name -> tensor) so the reference's functional layer API
(`scope=` strings) can be kept.

Two execution paths per layer:
  * is_training=False -> the BatchNorm is folded into (W, b) and the layer runs on
    the hand-written fp32-MFMA kernels of libpn2_hip.so (pn2_linear / fused SA);
  * is_training=True  -> differentiable torch ops (batch statistics cannot be
    fused through; the index/gather ops around it are still the HIP kernels).
"""
import contextlib
import math
from collections import OrderedDict

import weakref

import torch
import torch.nn.functional as F

from .._lib import PN2_EUNSUP as PN2_EUNSUP_CODE
from .._lib import check, lib, ptr, require_cuda, rows_in_place, stream_ptr

BN_EPSILON = 1e-3  # tf.contrib.layers.batch_norm default (tf_util.py:571-581)


class VariableStore:
    """name -> tensor registry standing in for TF's variable scopes."""

    def __init__(self, device="cuda", seed=0):
        self.device = torch.device(device)
        self.params = OrderedDict()   # trainable
        self.buffers = OrderedDict()  # moving averages
        self._gen = torch.Generator(device="cpu")
        self._gen.manual_seed(seed)
        self._seed = int(seed)
        self._dropout = {}
        self._folded = {}
        self.zero_arena = None  # ZeroArena of the training step (set by train.Trainer), None = every call zero-fills its own scratch
        # direct gradients (train.Trainer): parameter storage address -> (flat gradient buffer, offset, numel).  While
        # `grad_direct` is set -- inside a trainer's forward/backward, which zero-fills the flat buffer first -- the weight-gradient
        # and batch-norm kernels write each parameter's gradient straight into its slice (grad_view): no per-step pack copy
        self.grad_map = {}
        self.grad_direct = False
        self.train_epoch = 0  # bumped by every training-mode layer call: the HIP BN kernel updates the moving averages
                              # through raw pointers (no autograd version bump), so folded inference weights key on it too

    def grad_view(self, t):
        """a FRESH view, shaped like t, of the flat-gradient slice of the parameter whose storage t shares, or None (no trainer
        bound / outside its backward).  Fresh: autograd keeps a returned gradient without a copy only when nothing else holds it."""
        if not self.grad_direct:
            return None
        ent = self.grad_map.get(t.data_ptr())
        if ent is None or ent[2] != t.numel():
            return None
        flat, off, n = ent
        return flat[off:off + n].view(t.shape)

    def dropout_state(self, key, salt):
        """device int64[2] = {seed ^ salt, step} of one dropout call site; `set_step` advances every site."""
        t = self._dropout.get(key)
        if t is None:
            t = torch.tensor([(self._seed * 0x9E3779B1 + 12345) ^ int(salt), 0], dtype=torch.int64, device=self.device)
            self._dropout[key] = t
        return t

    def step_fills(self, step):
        """[(tensor, value)] that set_step would write: for a launch that stores them itself (tf_util.multi_copy_(fills=))"""
        return [(t[1:2], int(step)) for t in self._dropout.values()]

    def set_step(self, step):
        # a fill KERNEL on the current stream.  (`t[1] = int(step)` is a host-to-device copy of a pageable scalar: it blocks
        # the host until everything queued before it has run -- measured 4.2 ms per training step, tools/train_host_probe.py)
        for t in self._dropout.values():
            t[1:2].fill_(int(step))

    def get_variable(self, name, shape, init):
        if name not in self.params:
            self.params[name] = torch.nn.Parameter(init(shape).to(self.device))
        p = self.params[name]
        if tuple(p.shape) != tuple(shape):
            raise ValueError("variable %s exists with shape %s, requested %s" % (name, tuple(p.shape), tuple(shape)))
        return p

    def get_buffer(self, name, shape, value):
        if name not in self.buffers:
            self.buffers[name] = torch.full(shape, float(value), dtype=torch.float32, device=self.device)
        return self.buffers[name]

    def xavier(self, shape):
        # tf.contrib.layers.xavier_initializer(uniform=True): limit = sqrt(6 / (fan_in + fan_out))
        receptive = 1
        for d in shape[:-2]:
            receptive *= d
        fan_in, fan_out = shape[-2] * receptive, shape[-1] * receptive
        limit = math.sqrt(6.0 / (fan_in + fan_out))
        return (torch.rand(shape, generator=self._gen, dtype=torch.float32) * 2 - 1) * limit

    def parameters(self):
        return list(self.params.values())

    def num_parameters(self):
        return sum(p.numel() for p in self.params.values())

    def state_dict(self):
        d = OrderedDict((k, v.detach().clone()) for k, v in self.params.items())
        d.update((k, v.clone()) for k, v in self.buffers.items())
        return d

    def load_state_dict(self, state):
        with torch.no_grad():
            for k, v in state.items():
                tgt = self.params.get(k, self.buffers.get(k))
                if tgt is None:
                    raise KeyError(k)
                tgt.copy_(v.to(self.device))

    # folded inference weights, cached on the version counters of their sources
    def folded(self, key, sources, make):
        stamp = (self.train_epoch,) + tuple((id(t), t._version) for t in sources)
        hit = self._folded.get(key)
        if hit is None or hit[0] != stamp:
            with torch.no_grad():
                hit = (stamp, make())
            self._folded[key] = hit
        return hit[1]


class ZeroArena:
    """Per-step scratch that must START OUT ZERO (batch-norm accumulators, weight-gradient tiles that are added to with
    atomics).  `reset()` zero-fills the whole arena with ONE launch and rewinds it; `take(nbytes)` hands out consecutive
    256-byte aligned slices.  Replaces one memset per layer call (~110 per training step).  The first step runs
    without an arena and only measures (`needed`); the trainer then allocates it."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.buf = None
        self.off = 0
        self.needed = 0
        self._count = 0

    def allocate(self):
        self.buf = torch.zeros(max(256, self.needed + 256), dtype=torch.uint8, device=self.device)
        self.off = 0

    def reset(self):
        self._count = 0
        if self.buf is not None:
            self.buf.zero_()
            self.off = 0

    def take(self, nbytes):
        """-> a zero-filled uint8 view of nbytes, or None when the arena is not allocated yet / exhausted (the caller then
        uses the self-zeroing entry point)."""
        nbytes = (int(nbytes) + 255) & ~255
        self._count += nbytes
        if self._count > self.needed:
            self.needed = self._count
        if self.buf is None or self.off + nbytes > self.buf.numel():
            return None
        v = self.buf[self.off:self.off + nbytes]
        self.off += nbytes
        return v


_default_store = None
_scope_stack = []


def get_default_store():
    global _default_store
    if _default_store is None:
        _default_store = VariableStore()
    return _default_store


def set_default_store(store):
    global _default_store
    _default_store = store
    return store


@contextlib.contextmanager
def variable_scope(name):
    _scope_stack.append(name)
    try:
        yield "/".join(_scope_stack)
    finally:
        _scope_stack.pop()


def _full_name(leaf):
    return "/".join(_scope_stack + [leaf])


def _dense_variables(cin, cout, bn, kernel_shape):
    st = get_default_store()
    w = st.get_variable(_full_name("weights"), kernel_shape, st.xavier)
    b = st.get_variable(_full_name("biases"), (cout,), lambda s: torch.zeros(s))
    if not bn:
        return st, w, b, None
    with variable_scope("bn"):
        beta = st.get_variable(_full_name("beta"), (cout,), lambda s: torch.zeros(s))
        gamma = st.get_variable(_full_name("gamma"), (cout,), lambda s: torch.ones(s))
        mean = st.get_buffer(_full_name("moving_mean"), (cout,), 0.0)
        var = st.get_buffer(_full_name("moving_variance"), (cout,), 1.0)
    return st, w, b, (beta, gamma, mean, var)


def folded_dense(cin, cout, bn, kernel_shape, pad_to=None, pad_in=None, rotate_rows=0):
    """(W', b') with the inference BatchNorm folded in:
    y = (x@W + b - mean)/sqrt(var+eps)*gamma + beta = x@(W*s) + ((b-mean)*s + beta).
    rotate_rows = r moves the LAST r input rows of W' to the front (a kernel that feeds [xyz | features] to a
    layer whose variables were created for [features | xyz], pointnet_util.py:259)."""
    st, w, b, bnv = _dense_variables(cin, cout, bn, kernel_shape)
    key = (_full_name("folded"), pad_to, pad_in, rotate_rows)

    def make():
        w2 = w.detach().reshape(cin, cout)
        b2 = b.detach()
        if bnv is not None:
            beta, gamma, mean, var = bnv
            s = gamma.detach() / torch.sqrt(var + BN_EPSILON)
            w2 = w2 * s
            b2 = (b2 - mean) * s + beta.detach()
        if pad_to is not None and cout % pad_to != 0:
            padc = pad_to - cout % pad_to
            w2 = F.pad(w2, (0, padc))
            b2 = F.pad(b2, (0, padc))
        if rotate_rows:
            w2 = torch.cat([w2[cin - rotate_rows:], w2[:cin - rotate_rows]], dim=0)
        if pad_in is not None and pad_in > cin:  # zero rows for zero-padded input columns
            w2 = F.pad(w2, (0, 0, 0, pad_in - cin))
        return w2.contiguous(), b2.contiguous()

    srcs = [w, b] + (list(bnv) if bnv is not None else [])
    return st.folded(key, srcs, make)


def hip_linear(x2d, w, b, relu=True, pool=0):
    """y = relu?(x2d @ w + b), optional max over groups of `pool` consecutive rows.
    Thin wrapper over pn2_linear (fp32 MFMA)."""
    require_cuda(x2d, w, b)  # b may be None (no bias)
    rows, cin = x2d.shape
    cout = w.shape[1]
    x2d = x2d.contiguous()
    orows = rows // pool if pool and pool > 1 else rows
    y = torch.empty((orows, cout), dtype=torch.float32, device=x2d.device)
    with torch.cuda.device(x2d.device):
        check(lib.pn2_linear(rows, cin, cout, ptr(x2d), ptr(w), ptr(b), int(bool(relu)), int(pool or 0), ptr(y),
                             stream_ptr()), "pn2_linear")
    return y


def hip_mlp_chain(x2d, ws, bs, pool=0):
    """relu(relu(x @ W0 + b0) @ W1 + b1) with LDS-resident weights (pn2_mlp_chain).  Returns None
    when the library reports the configuration as unsupported (caller falls back to hip_linear)."""
    import ctypes
    from .._lib import PN2_EUNSUP
    require_cuda(x2d)
    rows, cin = x2d.shape
    x2d = x2d.contiguous()
    L = len(ws)
    widths = (ctypes.c_int * L)(*[w.shape[1] for w in ws])
    wptrs = (ctypes.c_void_p * L)(*[w.data_ptr() for w in ws])
    bptrs = (ctypes.c_void_p * L)(*[b.data_ptr() for b in bs])
    orows = rows // pool if pool else rows
    y = torch.empty((orows, ws[-1].shape[1]), dtype=torch.float32, device=x2d.device)
    with torch.cuda.device(x2d.device):
        rc = lib.pn2_mlp_chain(rows, cin, ptr(x2d), L, ctypes.cast(widths, ctypes.c_void_p),
                               ctypes.cast(wptrs, ctypes.c_void_p), ctypes.cast(bptrs, ctypes.c_void_p), int(pool),
                               ptr(y), stream_ptr())
    if rc == PN2_EUNSUP:
        return None
    check(rc, "pn2_mlp_chain")
    return y


def multi_copy_(dsts, srcs, fills=()):
    """dsts[i].copy_(srcs[i]) for lists of contiguous device tensors of any dtypes in ONE launch (pn2_multi_copy);
    torch._foreach_copy_ issues one memcpy per tensor when the dtypes are mixed.  fills = [(one-element float32 / int64 tensor,
    python value), ...] (<= 4): scalars stored by the same launch, their values travelling in the launch arguments
    (pn2_multi_copy_fill) -- a training step's lr_t and dropout step ride with its input copy."""
    import ctypes
    import struct
    n = len(dsts)
    if n != len(srcs):
        raise ValueError("multi_copy_: list lengths differ")
    for d, s in zip(dsts, srcs):
        if d.numel() * d.element_size() != s.numel() * s.element_size() or not d.is_contiguous() or not s.is_contiguous():
            raise ValueError("multi_copy_: contiguous tensors of equal byte size expected")
    from ..tf_ops.tf_sampling import drop_fps_tag
    for d in dsts:
        drop_fps_tag(d)  # a write through the raw pointer: no version counter moves, so the tie-record tag must not survive it
    fills = list(fills)
    if len(fills) > 4:
        raise ValueError("multi_copy_: at most 4 scalar fills")
    for i0 in range(0, n, 48):
        dd, ss = dsts[i0:i0 + 48], srcs[i0:i0 + 48]
        k = len(dd)
        sp = (ctypes.c_void_p * k)(*[t.data_ptr() for t in ss])
        dp = (ctypes.c_void_p * k)(*[t.data_ptr() for t in dd])
        nb = (ctypes.c_ulonglong * k)(*[t.numel() * t.element_size() for t in dd])
        with torch.cuda.device(dd[0].device):
            if fills and i0 == 0:
                nf = len(fills)
                vals, sizes = [], []
                for t, v in fills:
                    if t.numel() != 1 or t.dtype not in (torch.float32, torch.int64):
                        raise ValueError("multi_copy_: a fill target is one float32 or int64 element")
                    if t.dtype == torch.float32:
                        vals.append(struct.unpack("<I", struct.pack("<f", float(v)))[0])
                        sizes.append(4)
                    else:
                        vals.append(int(v) & 0xFFFFFFFFFFFFFFFF)
                        sizes.append(8)
                fp = (ctypes.c_void_p * nf)(*[t.data_ptr() for t, _ in fills])
                fv = (ctypes.c_ulonglong * nf)(*vals)
                fb = (ctypes.c_int * nf)(*sizes)
                check(lib.pn2_multi_copy_fill(k, ctypes.cast(sp, ctypes.c_void_p), ctypes.cast(dp, ctypes.c_void_p),
                                              ctypes.cast(nb, ctypes.c_void_p), nf, ctypes.cast(fp, ctypes.c_void_p),
                                              ctypes.cast(fv, ctypes.c_void_p), ctypes.cast(fb, ctypes.c_void_p), stream_ptr()),
                      "pn2_multi_copy_fill")
            else:
                check(lib.pn2_multi_copy(k, ctypes.cast(sp, ctypes.c_void_p), ctypes.cast(dp, ctypes.c_void_p),
                                         ctypes.cast(nb, ctypes.c_void_p), stream_ptr()), "pn2_multi_copy")


def _layer_arrays(ws, bs):
    import ctypes
    L = len(ws)
    widths = (ctypes.c_int * L)(*[w.shape[1] for w in ws])
    wptrs = (ctypes.c_void_p * L)(*[w.data_ptr() for w in ws])
    bptrs = (ctypes.c_void_p * L)(*[b.data_ptr() for b in bs])
    return L, ctypes.cast(widths, ctypes.c_void_p), ctypes.cast(wptrs, ctypes.c_void_p), ctypes.cast(bptrs, ctypes.c_void_p), (widths, wptrs, bptrs)


def hip_mlp_wide(x2d, ws, bs, relu_last=True, pool=0):
    """up to three dense layers of width 128 / 256 / 512 in ONE launch (pn2_mlp_wide: a workgroup carries a 32-row tile
    through all layers); pool = 32 adds the max over each group of 32 rows.  None when the library reports the
    configuration as unsupported (caller falls back to hip_linear per layer)."""
    from .._lib import PN2_EUNSUP
    require_cuda(x2d)
    x2d = x2d.contiguous()
    rows, cin = x2d.shape
    if ws[0].shape[0] < -(-cin // 8) * 8:  # the kernel reads W0 in groups of 8 rows
        ws = [F.pad(ws[0], (0, 0, 0, -(-cin // 8) * 8 - ws[0].shape[0])).contiguous()] + list(ws[1:])
    L, widths, wptrs, bptrs, keep = _layer_arrays(ws, bs)
    y = torch.empty((rows // pool if pool else rows, ws[-1].shape[1]), dtype=torch.float32, device=x2d.device)
    with torch.cuda.device(x2d.device):
        rc = lib.pn2_mlp_wide(rows, cin, cin, ptr(x2d), L, widths, wptrs, bptrs, int(bool(relu_last)), int(pool), ptr(y),
                              stream_ptr())
    if rc == PN2_EUNSUP:
        return None
    check(rc, "pn2_mlp_wide")
    return y


def hip_fp_mlp_wide(dist, idx, points1, points2, ws, bs):
    """[three_interpolate(points2) | points1] -> up to three wide dense layers in ONE launch (pn2_fp_mlp_wide);
    -> (b*n, w_last) or None when the library reports the configuration as unsupported."""
    from .._lib import PN2_EUNSUP
    require_cuda(dist, idx, points1, points2)
    b, n, _ = dist.shape
    m, c2 = points2.shape[1], points2.shape[2]
    c1 = 0 if points1 is None else points1.shape[2]
    need = -(-(c1 + c2) // 8) * 8
    if ws[0].shape[0] < need:
        ws = [F.pad(ws[0], (0, 0, 0, need - ws[0].shape[0])).contiguous()] + list(ws[1:])
    L, widths, wptrs, bptrs, keep = _layer_arrays(ws, bs)
    y = torch.empty((b * n, ws[-1].shape[1]), dtype=torch.float32, device=dist.device)
    p1 = None if points1 is None else points1.contiguous()
    with torch.cuda.device(dist.device):
        rc = lib.pn2_fp_mlp_wide(b, n, m, c1, c2, ptr(dist.contiguous()), ptr(idx.contiguous()), ptr(p1), ptr(points2.contiguous()),
                                 L, widths, wptrs, bptrs, ptr(y), stream_ptr())
    if rc == PN2_EUNSUP:
        return None
    check(rc, "pn2_fp_mlp_wide")
    return y


def sa_wide_first_layer(w0):
    """W0 (3 + c, cout) in the reference's [xyz | features] row order (pointnet_util.py:52-54) -> the row order
    pn2_sa_mlp_wide reads: [features | xyz | zero rows up to a multiple of 8]"""
    w = torch.cat([w0[3:], w0[:3]], dim=0)
    pad = -w.shape[0] % 8
    return F.pad(w, (0, 0, 0, pad)).contiguous() if pad else w.contiguous()


def hip_sa_mlp_wide(xyz, new_xyz, points, idx, ws, bs, pool=True):
    """SA front end (gather, centre, concat) + up to three wide dense layers + max over the 32 neighbours in ONE launch
    (pn2_sa_mlp_wide).  ws[0] in the kernel's row order (sa_wide_first_layer / folded_dense(rotate_rows, pad_in)).
    -> (b, m, w_last) (pool) or (b, m, 32, w_last); None when unsupported."""
    from .._lib import PN2_EUNSUP
    require_cuda(xyz, new_xyz, points, idx)
    b, n, _ = xyz.shape
    m, ns = idx.shape[1], idx.shape[2]
    c = points.shape[2]
    L, widths, wptrs, bptrs, keep = _layer_arrays(ws, bs)
    wl = ws[-1].shape[1]
    y = torch.empty((b, m, wl) if pool else (b, m, ns, wl), dtype=torch.float32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        rc = lib.pn2_sa_mlp_wide(b, n, m, ns, c, ptr(xyz.contiguous()), ptr(new_xyz.contiguous()), ptr(points.contiguous()),
                                 ptr(idx.contiguous()), L, widths, wptrs, bptrs, int(bool(pool)), ptr(y), stream_ptr())
    if rc == PN2_EUNSUP:
        return None
    check(rc, "pn2_sa_mlp_wide")
    return y


def hip_fp_mlp_wide_pre(dist, idx, points1, points2, ws, bs):
    """pn2_fp_mlp_wide with the first layer hoisted by linearity (see hip_fp_mlp_fused_pre): z = points2 @ W0[:c2] on the
    known points; the kernel's layer-0 tile holds only the skip-link channels.  -> (b*n, w_last) or None when unsupported."""
    from .._lib import PN2_EUNSUP
    require_cuda(dist, idx, points1, points2)
    if points1 is None:
        return None
    b, n, _ = dist.shape
    m, c2 = points2.shape[1], points2.shape[2]
    c1 = points1.shape[2]
    w0a, w0b = split_first_layer(ws[0], c2, c1, "fp_wide_pre", pad_b_rows=8)
    z = hoist_gemm(points2.reshape(b * m, c2), w0a)
    L, widths, wptrs, bptrs, keep = _layer_arrays([w0b] + list(ws[1:]), bs)
    y = torch.empty((b * n, ws[-1].shape[1]), dtype=torch.float32, device=dist.device)
    with torch.cuda.device(dist.device):
        rc = lib.pn2_fp_mlp_wide_pre(b, n, m, c1, ptr(dist.contiguous()), ptr(idx.contiguous()), ptr(points1.contiguous()), ptr(z),
                                     L, widths, wptrs, bptrs, ptr(y), stream_ptr())
    if rc == PN2_EUNSUP:
        return None
    check(rc, "pn2_fp_mlp_wide_pre")
    return y


def hip_sa_mlp_wide_pre(xyz, new_xyz, points, idx, ws, bs, pool=True):
    """pn2_sa_mlp_wide with the feature part of the first layer hoisted: zf = points @ W0[feature rows] on the source points.
    ws[0] in the wide kernel's row order [features (c) | xyz (3) | zero pad].  -> (b, m, w_last) or None when unsupported."""
    from .._lib import PN2_EUNSUP
    require_cuda(xyz, new_xyz, points, idx)
    b, n, _ = xyz.shape
    m, ns = idx.shape[1], idx.shape[2]
    c = points.shape[2]
    w0f, w0x = split_first_layer(ws[0], c, 3, "sa_wide_pre", pad_b_rows=8)
    zf = hoist_gemm(points.reshape(b * n, c), w0f)
    L, widths, wptrs, bptrs, keep = _layer_arrays([w0x] + list(ws[1:]), bs)
    wl = ws[-1].shape[1]
    y = torch.empty((b, m, wl) if pool else (b, m, ns, wl), dtype=torch.float32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        rc = lib.pn2_sa_mlp_wide_pre(b, n, m, ns, ptr(xyz.contiguous()), ptr(new_xyz.contiguous()), ptr(zf), ptr(idx.contiguous()),
                                     L, widths, wptrs, bptrs, int(bool(pool)), ptr(y), stream_ptr())
    if rc == PN2_EUNSUP:
        return None
    check(rc, "pn2_sa_mlp_wide_pre")
    return y


def hip_fp_mlp_fused(dist, idx, points1, points2, ws, bs):
    """[three_interpolate(points2) | points1] -> up to two dense layers in ONE kernel (pn2_fp_mlp_fused);
    returns (b*n, w_last) or None when the library reports the configuration as unsupported."""
    import ctypes
    from .._lib import PN2_EUNSUP
    require_cuda(dist, idx, points1, points2)
    b, n, _ = dist.shape
    m, c2 = points2.shape[1], points2.shape[2]
    c1 = 0 if points1 is None else points1.shape[2]
    p1 = None if points1 is None else points1.contiguous()
    p2 = points2.contiguous()
    L = len(ws)
    widths = (ctypes.c_int * L)(*[w.shape[1] for w in ws])
    wptrs = (ctypes.c_void_p * L)(*[w.data_ptr() for w in ws])
    bptrs = (ctypes.c_void_p * L)(*[bb.data_ptr() for bb in bs])
    y = torch.empty((b * n, ws[-1].shape[1]), dtype=torch.float32, device=dist.device)
    with torch.cuda.device(dist.device):
        rc = lib.pn2_fp_mlp_fused(b, n, m, c1, c2, ptr(dist.contiguous()), ptr(idx.contiguous()), ptr(p1), ptr(p2), L,
                                  ctypes.cast(widths, ctypes.c_void_p), ctypes.cast(wptrs, ctypes.c_void_p),
                                  ctypes.cast(bptrs, ctypes.c_void_p), ptr(y), stream_ptr())
    if rc == PN2_EUNSUP:
        return None
    check(rc, "pn2_fp_mlp_fused")
    return y


_zero_bias = {}
HOIST_GEMM_WIDE = False  # A/B: the hoisted products on pn2_mlp_wide instead of pn2_linear (measured: 17.4 vs 13.9 us at 4096 x 256 -> 256, 14.4 vs 13.3 at 16384 x 128 -> 128)


def hoist_gemm(x2d, w):
    """z = x2d @ w (no bias, no activation): the hoisted first-layer products.  4096 .. 65536 rows of width 128 / 256 / 512
    run on the wide-layer kernel (pn2_mlp_wide, single layer), everything else on pn2_linear."""
    rows, cin = x2d.shape
    cout = w.shape[1]
    if HOIST_GEMM_WIDE and 4096 <= rows <= 65536 and cout in (128, 256, 512) and cin % 8 == 0:
        key = (cout, str(w.device))
        zb = _zero_bias.get(key)
        if zb is None:
            zb = _zero_bias[key] = torch.zeros(cout, dtype=torch.float32, device=w.device)
        y = hip_mlp_wide(x2d.contiguous(), [w], [zb], relu_last=False)
        if y is not None:
            return y
    return hip_linear(x2d, w, None, relu=False)


def split_first_layer(w2, c2, c1, key, pad_b_rows=None):
    """(W1a, W1b) = rows [0, c2) and [c2, c2 + c1) of a folded first-layer weight (>= c2 + c1 rows, cout) as contiguous
    tensors (the part whose product is hoisted / the part that stays in the kernel); W1b zero-padded to a multiple of
    pad_b_rows rows when given.  Cached with the folded weight they come from."""
    st = get_default_store()

    def make():
        w1b = w2[c2:c2 + c1] if c1 > 0 else None
        if w1b is not None and pad_b_rows and c1 % pad_b_rows:
            w1b = F.pad(w1b, (0, 0, 0, pad_b_rows - c1 % pad_b_rows))
        return w2[:c2].contiguous(), (None if w1b is None else w1b.contiguous())
    return st.folded((_full_name("split"), key, c2, c1, pad_b_rows), [w2], make)


def hip_fp_mlp_fused_pre(dist, idx, points1, points2, ws, bs, schedule=None):
    """The FP block with the first layer's product hoisted out by linearity (pn2_fp_mlp_fused_pre):
    interp(points2) @ W1a == interp(points2 @ W1a), and Z = points2 @ W1a has the m KNOWN rows per cloud instead of the n
    unknown ones (8x fewer at every level of semantic.json).  Z is one pn2_linear call (no bias, no activation); the fused
    kernel blends three gathered rows of Z straight into its layer-1 accumulators and only the skip-link channels still
    go through the MFMA.  ws[0] = the folded first-layer weight ((c2 + c1 [+ pad]) x w1), 2 or 3 layers, all <= 128 wide.
    schedule (tests, A/B): 0 = the lockstep kernel, 1 = the software-pipelined one (pn2_fp_mlp_fused_pre_schedule); None = the
    library's choice.  Same bits either way.
    Returns (b*n, w_last), or None when the library reports the configuration as unsupported."""
    import ctypes
    from .._lib import PN2_EUNSUP
    require_cuda(dist, idx, points1, points2)
    b, n, _ = dist.shape
    m, c2 = points2.shape[1], points2.shape[2]
    c1 = 0 if points1 is None else points1.shape[2]
    L = len(ws)
    if L < 2 or L > 3 or any(w.shape[1] % 32 or w.shape[1] > 128 for w in ws):
        return None
    w1a, w1b = split_first_layer(ws[0], c2, c1, "fp_pre")
    z = hoist_gemm(points2.reshape(b * m, c2), w1a)  # (b*m, w1): the hoisted product
    # the skip link may be a column block of a wider batch (model.get_sa_fp_features: the rgb half of point_cloud): read in place
    p1, ld1 = (None, 0) if points1 is None else rows_in_place(points1)
    if schedule is not None and p1 is not None:
        p1, ld1 = p1.contiguous(), c1
    widths = (ctypes.c_int * L)(*[w.shape[1] for w in ws])
    wlist = [w1b] + list(ws[1:])
    wptrs = (ctypes.c_void_p * L)(*[(w.data_ptr() if w is not None else None) for w in wlist])
    bptrs = (ctypes.c_void_p * L)(*[bb.data_ptr() for bb in bs])
    y = torch.empty((b * n, ws[-1].shape[1]), dtype=torch.float32, device=dist.device)
    with torch.cuda.device(dist.device):
        common = (b, n, m, c1, ptr(dist.contiguous()), ptr(idx.contiguous()), ptr(p1), ptr(z), L,
                  ctypes.cast(widths, ctypes.c_void_p), ctypes.cast(wptrs, ctypes.c_void_p),
                  ctypes.cast(bptrs, ctypes.c_void_p), ptr(y))
        if schedule is None and p1 is not None and ld1 != c1:
            rc = lib.pn2_fp_mlp_fused_pre_ld(*common[:7], ld1, *common[7:], stream_ptr())
        elif schedule is None:
            rc = lib.pn2_fp_mlp_fused_pre(*common, stream_ptr())
        else:
            rc = lib.pn2_fp_mlp_fused_pre_schedule(*common, int(schedule), stream_ptr())
    if rc == PN2_EUNSUP:
        return None
    check(rc, "pn2_fp_mlp_fused_pre")
    return y


def hip_linear_narrow(x2d, w, b=None):
    """y = x2d @ w (+ b) for a layer of at most 16 outputs (the 9-class head) in ONE streaming launch (pn2_linear_narrow);
    None when the library reports the shape as unsupported (the caller pads to an MFMA tile)."""
    require_cuda(x2d, w, b)
    rows, cin = x2d.shape
    cout = w.shape[1]
    if cout > 16 or cin % 4 != 0 or x2d.dtype != torch.float32:
        return None
    x2d = x2d.contiguous()
    y = torch.empty((rows, cout), dtype=torch.float32, device=x2d.device)
    with torch.cuda.device(x2d.device):
        rc = lib.pn2_linear_narrow(rows, cin, cout, ptr(x2d), ptr(w.contiguous()), ptr(None if b is None else b.contiguous()), ptr(y),
                                   stream_ptr())
    if rc == PN2_EUNSUP_CODE:
        return None
    check(rc, "pn2_linear_narrow")
    return y


def hip_matmul(x2d, w):
    """y = x2d @ w on pn2_linear (no bias, no activation) -- the forward GEMM of the training path.  Output widths that
    are not a multiple of 32 run on pn2_linear_narrow (<= 16 outputs: the 9-class head) or with zero-padded weight columns,
    sliced back."""
    cin, cout = w.shape
    if cout <= 16:
        y = hip_linear_narrow(x2d, w)
        if y is not None:
            return y
    if cout % 32 != 0:
        wp = F.pad(w, (0, 32 - cout % 32)).contiguous()
        return hip_linear(x2d, wp, None, relu=False)[:, :cout].contiguous()
    return hip_linear(x2d, w.contiguous(), None, relu=False)


def hip_matmul_bn_stats(x2d, w, ws):
    """y = x2d @ w on pn2_linear_bn_stats: the GEMM also leaves the column sums of y and y^2 in the ZEROED batch-norm
    workspace `ws` (pn2_bn_workspace_bytes(cout) bytes) for pn2_bn_relu_forward_stats.  cout % 32 == 0."""
    require_cuda(x2d, w)
    rows, cin = x2d.shape
    cout = w.shape[1]
    y = torch.empty((rows, cout), dtype=torch.float32, device=x2d.device)
    with torch.cuda.device(x2d.device):
        check(lib.pn2_linear_bn_stats(rows, cin, cout, ptr(x2d.contiguous()), ptr(w.contiguous()), ptr(y), ptr(ws),
                                      ws.numel() * ws.element_size(), stream_ptr()), "pn2_linear_bn_stats")
    return y


def hip_matmul_bn_stats_xf(x_raw, w, ws, sc, sh, relu):
    """hip_matmul_bn_stats on relu?(fma(x_raw, sc, sh)) formed while the operand is staged (pn2_linear_bn_stats_xf)"""
    require_cuda(x_raw, w)
    rows, cin = x_raw.shape
    cout = w.shape[1]
    y = torch.empty((rows, cout), dtype=torch.float32, device=x_raw.device)
    with torch.cuda.device(x_raw.device):
        check(lib.pn2_linear_bn_stats_xf(rows, cin, cout, ptr(x_raw.contiguous()), ptr(w.contiguous()), ptr(y), ptr(ws),
                                         ws.numel() * ws.element_size(), ptr(sc), ptr(sh), int(relu), stream_ptr()),
              "pn2_linear_bn_stats_xf")
    return y


def hip_linear_dgrad(dy, w):
    """dx (rows, cin) = dy (rows, cout) @ w^T with w (cin, cout) as the forward pass holds it (pn2_linear_dgrad)."""
    require_cuda(dy, w)
    rows, cout = dy.shape
    cin = w.shape[0]
    dy = dy.contiguous()
    dx = torch.empty((rows, cin), dtype=torch.float32, device=dy.device)
    with torch.cuda.device(dy.device):
        check(lib.pn2_linear_dgrad(rows, cin, cout, ptr(dy), ptr(w.contiguous()), ptr(dx), stream_ptr()), "pn2_linear_dgrad")
    return dx


def _bn_scratch(c, device, fn, fn_ws0):
    """batch-norm accumulators: a slice of the step's zero arena (+ the entry point that trusts it) when there is one"""
    nbytes = lib.pn2_bn_workspace_bytes(c)
    arena = get_default_store().zero_arena
    v = arena.take(nbytes) if arena is not None else None
    if v is not None:
        return v, fn_ws0
    return torch.empty(nbytes // 8, dtype=torch.float64, device=device), fn


USE_GEMM_BN_STATS = True  # batch statistics from the forward GEMM's epilogue (set False: separate pass over y; tests / A-B)
# The one-block launches that used to follow every producer of batch-norm sums (fold the slot copies; derive scale / shift resp. the
# gradient constants) are done by the producer's LAST WORKGROUP (csrc/pn2_common.h pn2_bn_finish): a layer inside a stack is one
# launch forward (GEMM + statistics + constants) and two backward (data gradient + finish of the layer below; weight gradient).
# False: the separate launches (A/B, tests).
USE_BN_FINISH_IN_PRODUCER = True


def hip_matmul_bn_stats_fin(x2d, w, ws, xf, finish, gamma=None, beta=None, b=None, decay=0.0, running_mean=None, running_var=None):
    """hip_matmul_bn_stats (xf None) / hip_matmul_bn_stats_xf (xf = (scale, shift, relu)) whose last workgroup also folds the
    statistics (finish 1) or folds them and publishes the deferred batch norm's constants (finish 2, pn2_linear_bn_stats_fin).
    -> y, (save_mean, save_invstd, scale, shift) or None"""
    require_cuda(x2d, w)
    rows, cin = x2d.shape
    cout = w.shape[1]
    y = torch.empty((rows, cout), dtype=torch.float32, device=x2d.device)
    consts = None
    if finish == 2:
        save_mean = torch.empty(cout, dtype=torch.float32, device=x2d.device)
        consts = (save_mean, torch.empty_like(save_mean), torch.empty_like(save_mean), torch.empty_like(save_mean))
    sc, sh, xrelu = xf if xf is not None else (None, None, 0)
    cs = consts if consts is not None else (None, None, None, None)
    with torch.cuda.device(x2d.device):
        check(lib.pn2_linear_bn_stats_fin(rows, cin, cout, ptr(x2d.contiguous()), ptr(w.contiguous()), ptr(y), ptr(ws),
                                          ws.numel() * ws.element_size(), ptr(sc), ptr(sh), int(xrelu), int(finish), ptr(gamma),
                                          ptr(beta), ptr(b), BN_EPSILON, float(decay), ptr(running_mean), ptr(running_var),
                                          ptr(cs[0]), ptr(cs[1]), ptr(cs[2]), ptr(cs[3]), stream_ptr()), "pn2_linear_bn_stats_fin")
    return y, consts


def _bn_zeroed_scratch(c, device):
    """zero-filled batch-norm accumulators for pn2_linear_bn_stats: a slice of the step's zero arena, else a fresh fill"""
    nbytes = lib.pn2_bn_workspace_bytes(c)
    arena = get_default_store().zero_arena
    v = arena.take(nbytes) if arena is not None else None
    return v if v is not None else torch.zeros(nbytes // 8, dtype=torch.float64, device=device)


def _hip_wgrad(x2d, dy, w, xf=None):
    """dW = x2d^T @ dy on pn2_linear_wgrad; the tile is added to with atomics, so it starts from the zero arena when the
    step has one (no memset of its own).  xf = (scale, shift, relu): x2d is the producer's un-normalised output, the
    kernel applies its batch norm while loading (pn2_linear_wgrad_accumulate_xf)."""
    gv = get_default_store().grad_view(w)  # the parameter's slice of the trainer's (zero-filled) flat gradient
    if gv is not None:
        with torch.cuda.device(w.device):
            if xf is not None:
                check(lib.pn2_linear_wgrad_accumulate_xf(x2d.shape[0], w.shape[0], w.shape[1], ptr(x2d), ptr(dy), ptr(gv), ptr(xf[0]),
                                                         ptr(xf[1]), int(xf[2]), stream_ptr()), "pn2_linear_wgrad_accumulate_xf")
            else:
                check(lib.pn2_linear_wgrad_accumulate(x2d.shape[0], w.shape[0], w.shape[1], ptr(x2d), ptr(dy), ptr(gv), stream_ptr()),
                      "pn2_linear_wgrad_accumulate")
        return gv
    arena = get_default_store().zero_arena
    v = arena.take(w.numel() * 4) if arena is not None else None
    with torch.cuda.device(w.device):
        if xf is not None:
            dw = v[:w.numel() * 4].view(torch.float32).view_as(w) if v is not None else torch.zeros_like(w)
            check(lib.pn2_linear_wgrad_accumulate_xf(x2d.shape[0], w.shape[0], w.shape[1], ptr(x2d), ptr(dy), ptr(dw), ptr(xf[0]),
                                                     ptr(xf[1]), int(xf[2]), stream_ptr()), "pn2_linear_wgrad_accumulate_xf")
        elif v is not None:
            dw = v[:w.numel() * 4].view(torch.float32).view_as(w)
            check(lib.pn2_linear_wgrad_accumulate(x2d.shape[0], w.shape[0], w.shape[1], ptr(x2d), ptr(dy), ptr(dw), stream_ptr()),
                  "pn2_linear_wgrad_accumulate")
        else:
            dw = torch.empty_like(w)
            check(lib.pn2_linear_wgrad(x2d.shape[0], w.shape[0], w.shape[1], ptr(x2d), ptr(dy), ptr(dw), stream_ptr()),
                  "pn2_linear_wgrad")
    return dw


_ones_cache = {}


def _ones_column(rows, device):
    """(rows, 1) of ones: x operand that turns pn2_linear_wgrad into a column sum (bias gradient)"""
    key = (rows, str(device))
    t = _ones_cache.get(key)
    if t is None:
        _ones_cache.clear()  # one size at a time is enough (the head of the network)
        t = _ones_cache[key] = torch.ones((rows, 1), dtype=torch.float32, device=device)
    return t


class _TrainMatmul(torch.autograd.Function):
    """y = x2d @ w (+ b) for the training path's un-normalised layers (the class head); backward: dX on
    pn2_linear_dgrad, dW = x2d^T @ dY on pn2_linear_wgrad (the reduction over all rows, ~8x faster than the library
    GEMM on these tall-skinny shapes), db = 1^T @ dY on the same kernel (a column sum)."""

    @staticmethod
    def forward(ctx, x2d, w, b=None):
        ctx.save_for_backward(x2d, w, *(() if b is None else (b,)))
        ctx.has_bias = b is not None
        if w.shape[1] <= 16:  # the class head: GEMM + bias in one streaming launch
            y = hip_linear_narrow(x2d, w, b)
            if y is not None:
                return y
        y = hip_matmul(x2d, w)
        return y if b is None else y.add_(b)

    @staticmethod
    def backward(ctx, dy):
        x2d, w = ctx.saved_tensors[:2]
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = hip_linear_dgrad(dy, w)
        dw = None
        if ctx.needs_input_grad[1]:
            dw = _hip_wgrad(x2d, dy, w)
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _hip_wgrad(_ones_column(dy.shape[0], dy.device), dy, ctx.saved_tensors[2].view(1, -1)).reshape(-1)
        return dx, dw, db


class _TrainDenseRelu(torch.autograd.Function):
    """z = relu(x2d @ w + b): the training path of a layer WITHOUT batch norm (reference tf_util.py:186-204 with bn=False:
    conv -> bias_add -> relu).  Forward: pn2_linear with its bias + ReLU epilogue; backward: pn2_relu_grad (TF's ReluGrad: the
    upstream gradient where the OUTPUT is positive), then the data / weight / bias gradients of _TrainMatmul."""

    @staticmethod
    def forward(ctx, x2d, w, b):
        cin, cout = w.shape
        if cout % 32 != 0:  # pn2_linear's output tiles are 32 wide: zero-padded weight / bias columns, sliced back
            pad = 32 - cout % 32
            z = hip_linear(x2d, F.pad(w, (0, pad)).contiguous(), None if b is None else F.pad(b, (0, pad)).contiguous(),
                           relu=True)[:, :cout].contiguous()
        else:
            z = hip_linear(x2d, w.contiguous(), None if b is None else b.contiguous(), relu=True)
        ctx.save_for_backward(x2d, w, z)
        ctx.has_bias = b is not None
        return z

    @staticmethod
    def backward(ctx, dz):
        x2d, w, z = ctx.saved_tensors
        dz = dz.contiguous()
        dy = torch.empty_like(dz)
        with torch.cuda.device(dz.device):
            check(lib.pn2_relu_grad(dz.numel(), ptr(z), ptr(dz), ptr(dy), stream_ptr()), "pn2_relu_grad")
        dx = hip_linear_dgrad(dy, w) if ctx.needs_input_grad[0] else None
        dw = _hip_wgrad(x2d, dy, w) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _hip_wgrad(_ones_column(dy.shape[0], dy.device), dy, dy.new_empty((1, dy.shape[1]))).reshape(-1)
        return dx, dw, db


def _train_dense(inputs, w2d, b, relu=False):
    """un-normalised dense layer of the training path (the class head): GEMM, data and weight gradients on the HIP library"""
    cin, cout = w2d.shape
    require_cuda(inputs)
    if inputs.dtype != torch.float32:
        raise TypeError("the training path is float32")
    fn = _TrainDenseRelu if relu else _TrainMatmul
    y = fn.apply(inputs.reshape(-1, cin).contiguous(), w2d.contiguous(), b)
    return y.reshape(list(inputs.shape[:-1]) + [cout])


# ---- cross-layer link: the first batch-norm reduction of layer i rides in the data-gradient GEMM of layer i+1 -----------
# When the output z of a dense+BN(+ReLU) layer is consumed by exactly one other such layer, the gradient dz reaching the
# batch norm is the dX of that consumer's data-gradient GEMM.  pn2_linear_dgrad_bn_grad_stats forms the two per-channel
# sums of the batch-norm gradient (sum g, sum g * xhat) from its accumulator tiles, so the producer's backward skips the
# reduction pass over (dz, y) (pn2_bn_relu_backward_stats).  The link is keyed by the storage address of z and checked
# again in the backward: only when the tensor arriving as dz IS the linked GEMM's output (a second consumer makes autograd
# hand over a sum in a new tensor) is the shortcut taken.
USE_DGRAD_BN_STATS = True
# Registry of the producer records, keyed by the storage address of the producer's output.  WEAK values: the record is owned by
# the producer's autograd context (ctx.link), so it lives exactly as long as that node of the tape -- a forward pass whose tape
# is dropped (or never built) leaves nothing behind, and an address reused by a later tensor finds no stale record.
_bn_links = weakref.WeakValueDictionary()


class _BnLink:
    __slots__ = ("shape", "y", "gamma", "beta", "mean", "invstd", "relu", "ws", "dz_ptr", "dz_keep", "sc", "sh", "consumed",
                 "gx", "folded", "coef", "dgamma", "dbeta", "__weakref__")


# ---- deferred normalisation: layer i publishes (scale, shift) and hands its PRE-normalisation output y to layer i+1 -------
# (pn2_bn_relu_forward_deferred); layer i+1's forward GEMM and weight gradient apply relu(fma(y, scale, shift)) while they
# load y (pn2_linear_bn_stats_xf, pn2_linear_wgrad_accumulate_xf): the normalised activation is never written or re-read.
# Decided by the module code (conv2d(..., defer_bn=True)) for layers whose output feeds exactly the next conv2d of the stack.
USE_BN_ON_LOAD = True


def can_defer_bn(rows, c, c_next):
    """may a layer of width c over `rows` rows hand its un-normalised output to a following layer of width c_next?
    (Only while a tape is being recorded: the record that tells the next layer what it is reading belongs to the tape.)"""
    return bool(USE_BN_ON_LOAD and USE_DGRAD_BN_STATS and torch.is_grad_enabled() and rows > 2048 and c % 4 == 0
                and c_next % 32 == 0)


def assert_not_deferred(x, what):
    """raise if x is the UN-normalised output of a layer that deferred its batch norm (conv2d(..., defer_bn=True)): only the
    next dense layer of the stack may read such a tensor -- anything else would silently compute on un-normalised values"""
    if x is None or not _bn_links:
        return
    lk = _bn_links.get(x.data_ptr())
    if lk is not None and lk.sc is not None and not lk.consumed and lk.shape == (x.numel() // x.shape[-1], x.shape[-1]):
        raise RuntimeError("a deferred batch-norm output (conv2d(..., defer_bn=True)) reached %s" % what)


def _deferred_producer(x2d):
    lk = _bn_links.get(x2d.data_ptr())
    return lk if (lk is not None and lk.sc is not None and lk.y is not None and lk.shape == tuple(x2d.shape)) else None


def reset_bn_links():
    """forget the producer records of the previous forward pass (model.get_model calls this when training)"""
    _bn_links.clear()


def hip_linear_dgrad_linked(dy, w, link):
    """hip_linear_dgrad whose result is the gradient reaching the batch norm of the layer recorded in `link`: the GEMM also
    leaves that batch norm's two gradient sums in a zeroed workspace (pn2_linear_dgrad_bn_grad_stats) and notes both on the
    record for the producer's backward."""
    if USE_BN_FINISH_IN_PRODUCER and w.shape[1] > 16:
        return _hip_dgrad_fin(dy.contiguous(), None, w, link)
    rows, cin = dy.shape[0], w.shape[0]
    pws = _bn_zeroed_scratch(cin, dy.device)
    dx = torch.empty((rows, cin), dtype=torch.float32, device=dy.device)
    with torch.cuda.device(dy.device):
        check(lib.pn2_linear_dgrad_bn_grad_stats(rows, cin, w.shape[1], ptr(dy), ptr(w.contiguous()), ptr(dx), ptr(link.y),
                                                 ptr(link.gamma), ptr(link.beta), ptr(link.mean), ptr(link.invstd),
                                                 int(link.relu), ptr(pws), pws.numel() * pws.element_size(), stream_ptr()),
              "pn2_linear_dgrad_bn_grad_stats")
    link.ws, link.dz_ptr, link.dz_keep = pws, dx.data_ptr(), dx
    return dx


def _bn_train_forward(y, b, gamma, beta, running_mean, running_var, decay, relu, pool, stats_ws=None, folded=False):
    """batch norm (+ReLU, + max over groups of `pool` rows) of a layer output y (rows, c) on pn2_bn_relu_forward; stats_ws: the
    workspace pn2_linear_bn_stats has already left the column sums in.  -> z, ties, save_mean, save_invstd"""
    rows, c = y.shape
    if stats_ws is not None:
        ws, fwd = stats_ws, lib.pn2_bn_relu_forward_stats
    else:
        ws, fwd = _bn_scratch(c, y.device, lib.pn2_bn_relu_forward, lib.pn2_bn_relu_forward_ws0)
    pooled = pool > 1
    z = torch.empty((rows // pool, c) if pooled else (rows, c), dtype=y.dtype, device=y.device)
    save_mean = torch.empty(c, dtype=torch.float32, device=y.device)
    save_invstd = torch.empty_like(save_mean)
    ties = None
    with torch.cuda.device(y.device):
        if pooled:
            # ties = [tie counts | ysel]: ysel = the pre-normalisation value of the first row attaining each maximum, which lets the
            # backward take its reduction from the pooled tensors (pn2_bn_grad_constants) instead of a pass over y
            ties = torch.empty((2,) + tuple(z.shape), dtype=y.dtype, device=y.device)
            mode = (3 if folded else 2) if stats_ws is not None else (1 if fwd is lib.pn2_bn_relu_forward_ws0 else 0)
            check(lib.pn2_bn_relu_forward_pool(rows, c, ptr(y), ptr(gamma), ptr(beta), ptr(b), BN_EPSILON, decay, int(relu), int(pool),
                                               ptr(running_mean), ptr(running_var), ptr(ws), ws.numel() * ws.element_size(), mode,
                                               ptr(save_mean), ptr(save_invstd), ptr(z), ptr(ties[0]), ptr(ties[1]), stream_ptr()),
                  "pn2_bn_relu_forward_pool")
        elif folded and stats_ws is not None:  # the GEMM's last workgroup has folded the sums: only the normalisation pass is left
            check(lib.pn2_bn_relu_forward_mode(rows, c, ptr(y), ptr(gamma), ptr(beta), ptr(b), BN_EPSILON, decay, int(relu),
                                               ptr(running_mean), ptr(running_var), ptr(ws), ws.numel() * ws.element_size(), 3,
                                               ptr(save_mean), ptr(save_invstd), ptr(z), stream_ptr()), "pn2_bn_relu_forward_mode")
        else:
            check(fwd(rows, c, ptr(y), ptr(gamma), ptr(beta), ptr(b), BN_EPSILON, decay, int(relu),
                      int(pool), ptr(running_mean), ptr(running_var), ptr(ws), ws.numel() * ws.element_size(),
                      ptr(save_mean), ptr(save_invstd), ptr(z), None, stream_ptr()),
                  "pn2_bn_relu_forward")
    return z, ties, save_mean, save_invstd


def _bn_train_forward_deferred(y, b, gamma, beta, running_mean, running_var, decay, stats_ws=None):
    """the statistics half of _bn_train_forward only (pn2_bn_relu_forward_deferred) -> save_mean, save_invstd, scale, shift"""
    rows, c = y.shape
    ws = stats_ws if stats_ws is not None else _bn_zeroed_scratch(c, y.device)
    save_mean = torch.empty(c, dtype=torch.float32, device=y.device)
    save_invstd, sc, sh = torch.empty_like(save_mean), torch.empty_like(save_mean), torch.empty_like(save_mean)
    with torch.cuda.device(y.device):
        check(lib.pn2_bn_relu_forward_deferred(rows, c, ptr(y), ptr(gamma), ptr(beta), ptr(b), BN_EPSILON, decay,
                                               int(stats_ws is not None), ptr(running_mean), ptr(running_var), ptr(ws),
                                               ws.numel() * ws.element_size(), ptr(save_mean), ptr(save_invstd), ptr(sc), ptr(sh),
                                               stream_ptr()), "pn2_bn_relu_forward_deferred")
    return save_mean, save_invstd, sc, sh


def _bn_register_producer(z, y, gamma, beta, save_mean, save_invstd, relu, pooled, sc=None, sh=None, gx=False):
    """record an un-pooled dense+BN layer as the possible producer of the next layer's input (see _BnLink) -> link or None;
    sc / sh: the layer deferred its normalisation, z IS y"""
    if not USE_DGRAD_BN_STATS or pooled:
        return None
    lk = _BnLink()
    lk.shape, lk.y, lk.gamma, lk.beta, lk.mean, lk.invstd, lk.relu = tuple(z.shape), y, gamma, beta, save_mean, save_invstd, bool(relu)
    lk.ws = lk.dz_ptr = lk.dz_keep = None
    lk.sc, lk.sh, lk.consumed = sc, sh, False
    # gx: this layer's backward forms its batch-norm gradient on load -- the consumer's data-gradient GEMM may then publish the
    # gradient constants itself (finish kind 3); folded / coef / dgamma / dbeta: what that GEMM's last workgroup has left
    lk.gx, lk.folded, lk.coef, lk.dgamma, lk.dbeta = bool(gx), False, None, None, None
    _bn_links[z.data_ptr()] = lk
    return lk


def _param_grad_out(p):
    """where a kernel WRITES the gradient of parameter p: its slice of the trainer's flat gradient, else a fresh tensor"""
    gv = get_default_store().grad_view(p)
    return gv if gv is not None else torch.empty_like(p)


def _bn_train_backward(dz, y, gamma, beta, save_mean, save_invstd, relu, pool, zmax, ties, lk):
    """gradient of _bn_train_forward on pn2_bn_relu_backward -> dy (rows, c), dgamma, dbeta.  lk: this layer's producer
    record; when the consumer's data-gradient GEMM has already left the two reduction sums there (and dz is that GEMM's
    output), the reduction pass is skipped."""
    rows, c = y.shape
    dz = dz.contiguous()
    dy = torch.empty_like(y)
    dgamma, dbeta = _param_grad_out(gamma), _param_grad_out(beta)
    mode3 = False
    if lk is not None and lk.ws is not None and lk.dz_ptr == dz.data_ptr() and dz.shape == y.shape:
        ws, bwd, mode3 = lk.ws, lib.pn2_bn_relu_backward_stats, bool(lk.folded)
    else:
        ws, bwd = _bn_scratch(c, y.device, lib.pn2_bn_relu_backward, lib.pn2_bn_relu_backward_ws0)
    if lk is not None and lk.sc is not None and not lk.consumed:
        # the un-normalised output went somewhere else than into the next dense layer: whatever read it saw wrong values
        raise RuntimeError("a deferred batch-norm output (conv2d(..., defer_bn=True)) was not consumed by a following conv2d")
    if lk is not None:  # this layer's backward runs once: drop what the record kept alive
        lk.ws = lk.dz_keep = lk.dz_ptr = lk.y = lk.gamma = lk.beta = lk.mean = lk.invstd = lk.sc = lk.sh = None
        lk.coef = lk.dgamma = lk.dbeta = None
    if ties is not None and ties.dim() == zmax.dim() + 1:
        ties = ties[0]  # [tie counts | ysel] of _bn_train_forward
    with torch.cuda.device(y.device):
        if mode3:  # the consumer's data-gradient GEMM has left the sums AND folded them (pn2_linear_dgrad_fin)
            check(lib.pn2_bn_relu_backward_mode(rows, c, ptr(dz), ptr(y), ptr(gamma), ptr(beta), ptr(save_mean), ptr(save_invstd),
                                                int(relu), int(pool), ptr(zmax), ptr(ties), ptr(ws), ws.numel() * ws.element_size(),
                                                3, ptr(dy), ptr(dgamma), ptr(dbeta), stream_ptr()), "pn2_bn_relu_backward_mode")
        else:
            check(bwd(rows, c, ptr(dz), ptr(y), ptr(gamma), ptr(beta), ptr(save_mean),
                      ptr(save_invstd), int(relu), int(pool), ptr(zmax), ptr(ties), ptr(ws),
                      ws.numel() * ws.element_size(), ptr(dy), ptr(dgamma), ptr(dbeta), stream_ptr()),
                  "pn2_bn_relu_backward")
    return dy, dgamma, dbeta


# ---- batch-norm gradient applied on load (round 6): the gradient dy LEAVING a layer's batch norm is only ever read by that
# layer's own data and weight gradient GEMMs, so it is not written: pn2_bn_grad_constants folds the two reduction sums into six
# per-channel constants (+ dgamma, dbeta) and pn2_linear_dgrad_gx / pn2_linear_wgrad_gx form dy from (y, dz) while they load
# their operand -- bn_grad_apply_kernel's pass (read dz, y; write dy) and the two re-reads of dy are gone, and behind the fused
# max pool the (rows, c) gradient never exists at all.  Same float expressions as the materialised form: both hand the GEMMs the
# same bits (tests/test_train_gpu.py::test_bn_grad_on_load_equals_the_materialised_form).
USE_BN_GRAD_ON_LOAD = True
# behind the fused max pool the first backward reduction is taken from the pooled tensors (rows / 32 entries per channel: the
# gradient is non-zero only on the rows attaining a maximum) instead of a pass over y (A/B, tests)
USE_POOLED_BN_REDUCE = True


def _gx_usable(rows, c, pool, dz, y):
    return bool(USE_BN_GRAD_ON_LOAD and c % 4 == 0 and 16 < c <= 512 and pool in (0, 32) and (not pool or rows % 32 == 0)
                and dz.dtype == torch.float32 and dz.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0)


def _bn_grad_constants(dz, y, gamma, beta, save_mean, save_invstd, relu, pool, zmax, ties, lk):
    """first half of the on-load batch-norm gradient (pn2_bn_grad_constants) -> coef (6, c), dgamma, dbeta.  lk as in
    _bn_train_backward: the reduction pass is skipped when the consumer's data-gradient GEMM has already left the sums."""
    rows, c = y.shape
    ready = None
    if lk is not None and lk.ws is not None and lk.dz_ptr == dz.data_ptr() and dz.shape == y.shape:
        ws, done = lk.ws, 1
        if lk.coef is not None:  # the consumer's data-gradient GEMM has published them already (finish kind 3)
            ready = (lk.coef, lk.dgamma, lk.dbeta)
    else:
        ws, done = _bn_zeroed_scratch(c, y.device), 0
    if lk is not None and lk.sc is not None and not lk.consumed:
        raise RuntimeError("a deferred batch-norm output (conv2d(..., defer_bn=True)) was not consumed by a following conv2d")
    if lk is not None:
        lk.ws = lk.dz_keep = lk.dz_ptr = lk.y = lk.gamma = lk.beta = lk.mean = lk.invstd = lk.sc = lk.sh = None
        lk.coef = lk.dgamma = lk.dbeta = None
    if ready is not None:
        return ready
    coef = torch.empty((6, c), dtype=torch.float32, device=y.device)
    dgamma, dbeta = _param_grad_out(gamma), _param_grad_out(beta)
    ysel = None
    if ties is not None and ties.dim() == zmax.dim() + 1:
        ties, ysel = ties[0], (ties[1] if USE_POOLED_BN_REDUCE else None)
    with torch.cuda.device(y.device):
        check(lib.pn2_bn_grad_constants(rows, c, ptr(dz), ptr(y), ptr(gamma), ptr(beta), ptr(save_mean), ptr(save_invstd), int(relu),
                                        int(pool), ptr(zmax), ptr(ties), ptr(ysel), done, ptr(ws), ws.numel() * ws.element_size(),
                                        ptr(coef), ptr(dgamma), ptr(dbeta), stream_ptr()), "pn2_bn_grad_constants")
    return coef, dgamma, dbeta


def _hip_dgrad_fin(dy, gx, w, link):
    """The data gradient dx = dy @ w^T of a dense layer with everything the training path hangs on it (pn2_linear_dgrad_fin):
    dy given, or gx = (y, dz, coef, relu, pool, zmax, ties): formed on load; link: the producer record of the layer below -- its
    two batch-norm gradient sums come from the accumulator tiles, and the GEMM's last workgroup folds them (link.gx False: the
    materialised backward follows) or publishes that layer's gradient constants, dgamma and dbeta (link.gx True)."""
    ref = dy if dy is not None else gx[0]
    rows, cin = ref.shape[0], w.shape[0]
    dx = torch.empty((rows, cin), dtype=torch.float32, device=ref.device)
    below, fin_out = (None, None, None, None, None, 0, None, 0), (0, None, None, None)
    pws = None
    if link is not None:
        pws = _bn_zeroed_scratch(cin, ref.device)
        below = (ptr(link.y), ptr(link.gamma), ptr(link.beta), ptr(link.mean), ptr(link.invstd), int(link.relu), ptr(pws),
                 pws.numel() * pws.element_size())
        if link.gx and _gx_usable(rows, cin, 0, dx, link.y):
            link.coef = torch.empty((6, cin), dtype=torch.float32, device=ref.device)
            link.dgamma, link.dbeta = _param_grad_out(link.gamma), _param_grad_out(link.beta)
            fin_out = (3, ptr(link.coef), ptr(link.dgamma), ptr(link.dbeta))
        else:
            fin_out = (1, None, None, None)
    if gx is not None:
        y, dz, coef, relu, pool, zmax, ties = gx
        up = (None, ptr(y), ptr(dz), ptr(coef), int(relu), int(pool), ptr(zmax), ptr(ties))
    else:
        up = (ptr(dy), None, None, None, 0, 0, None, None)
    with torch.cuda.device(ref.device):
        check(lib.pn2_linear_dgrad_fin(rows, cin, w.shape[1], *up, ptr(w.contiguous()), ptr(dx), *below, *fin_out, stream_ptr()),
              "pn2_linear_dgrad_fin")
    if link is not None:
        link.ws, link.dz_ptr, link.dz_keep, link.folded = pws, dx.data_ptr(), dx, True
    return dx


USE_FUSED_BWD_NARROW = True  # 32 / 64-channel layers: data and weight gradient in one launch, (y, dz) read once (A/B, tests)


def _hip_bwd_fused(x2d, xf, y, dz, coef, relu, pool, zmax, ties, w, link):
    """dx AND dW of a narrow layer in ONE launch (pn2_linear_bwd_fused): what _hip_dgrad_fin + _hip_wgrad_gx return, or None when
    the library reports the shape as unsupported (the caller then launches the two)."""
    rows, cin, cout = y.shape[0], w.shape[0], w.shape[1]
    if not (USE_FUSED_BWD_NARROW and USE_BN_FINISH_IN_PRODUCER and cin in (32, 64) and cout in (32, 64) and rows % 32 == 0
            and x2d.is_contiguous() and x2d.shape[1] == cin):
        return None
    dx = torch.empty((rows, cin), dtype=torch.float32, device=y.device)
    dw = get_default_store().grad_view(w)
    if dw is None:
        arena = get_default_store().zero_arena
        v = arena.take(w.numel() * 4) if arena is not None else None
        dw = v[:w.numel() * 4].view(torch.float32).view_as(w) if v is not None else torch.zeros_like(w)
    below, fin_out, pws = (None, None, None, None, None, 0, None, 0), (0, None, None, None), None
    gxb = False
    if link is not None:
        pws = _bn_zeroed_scratch(cin, y.device)
        below = (ptr(link.y), ptr(link.gamma), ptr(link.beta), ptr(link.mean), ptr(link.invstd), int(link.relu), ptr(pws),
                 pws.numel() * pws.element_size())
        gxb = bool(link.gx and _gx_usable(rows, cin, 0, dx, link.y))
        if gxb:
            cb = torch.empty((6, cin), dtype=torch.float32, device=y.device)
            dgb, dbb = _param_grad_out(link.gamma), _param_grad_out(link.beta)
            fin_out = (3, ptr(cb), ptr(dgb), ptr(dbb))
        else:
            fin_out = (1, None, None, None)
    sc, sh, xrelu = xf if xf is not None else (None, None, 0)
    with torch.cuda.device(y.device):
        rc = lib.pn2_linear_bwd_fused(rows, cin, cout, ptr(x2d), ptr(sc), ptr(sh), int(xrelu), ptr(y), ptr(dz), ptr(coef), int(relu),
                                      int(pool), ptr(zmax), ptr(ties), ptr(w.contiguous()), ptr(dx), ptr(dw), *below, *fin_out,
                                      stream_ptr())
    if rc == PN2_EUNSUP_CODE:
        return None
    check(rc, "pn2_linear_bwd_fused")
    if link is not None:
        if gxb:
            link.coef, link.dgamma, link.dbeta = cb, dgb, dbb
        link.ws, link.dz_ptr, link.dz_keep, link.folded = pws, dx.data_ptr(), dx, True
    return dx, dw


def _hip_dgrad_gx(y, dz, coef, relu, pool, zmax, ties, w, link):
    """dx = dy @ w^T with dy formed on load (pn2_linear_dgrad_gx); link: the producer record of the layer below (its two batch-norm
    gradient sums are left in a zeroed workspace and noted on the record, as hip_linear_dgrad_linked does) or None"""
    if USE_BN_FINISH_IN_PRODUCER:
        return _hip_dgrad_fin(None, (y, dz, coef, relu, pool, zmax, ties), w, link)
    rows, cin = y.shape[0], w.shape[0]
    dx = torch.empty((rows, cin), dtype=torch.float32, device=y.device)
    pws = _bn_zeroed_scratch(cin, y.device) if link is not None else None
    with torch.cuda.device(y.device):
        if link is not None:
            below = (ptr(link.y), ptr(link.gamma), ptr(link.beta), ptr(link.mean), ptr(link.invstd), int(link.relu), ptr(pws),
                     pws.numel() * pws.element_size())
        else:
            below = (None, None, None, None, None, 0, None, 0)
        check(lib.pn2_linear_dgrad_gx(rows, cin, w.shape[1], ptr(y), ptr(dz), ptr(coef), int(relu), int(pool), ptr(zmax), ptr(ties),
                                      ptr(w.contiguous()), ptr(dx), *below, stream_ptr()), "pn2_linear_dgrad_gx")
    if link is not None:
        link.ws, link.dz_ptr, link.dz_keep = pws, dx.data_ptr(), dx
    return dx


def _hip_wgrad_gx(x2d, xf, y, dz, coef, relu, pool, zmax, ties, w):
    """dW = x2d^T @ dy with dy formed on load (pn2_linear_wgrad_gx); xf as in _hip_wgrad"""
    dw = get_default_store().grad_view(w)
    if dw is None:
        arena = get_default_store().zero_arena
        v = arena.take(w.numel() * 4) if arena is not None else None
        dw = v[:w.numel() * 4].view(torch.float32).view_as(w) if v is not None else torch.zeros_like(w)
    sc, sh, xrelu = xf if xf is not None else (None, None, 0)
    with torch.cuda.device(w.device):
        check(lib.pn2_linear_wgrad_gx(x2d.shape[0], w.shape[0], w.shape[1], ptr(x2d), ptr(sc), ptr(sh), int(xrelu), ptr(y), ptr(dz),
                                      ptr(coef), int(relu), int(pool), ptr(zmax), ptr(ties), ptr(dw), stream_ptr()),
              "pn2_linear_wgrad_gx")
    return dw


class _TrainDenseBnRelu(torch.autograd.Function):
    """relu?(batch_norm(x2d @ w + b)) [-> max over groups of `pool` rows] for the training path, with the
    normalisation on the HIP library: forward = GEMM -> pn2_bn_relu_forward (fp64 batch moments, normalise + ReLU
    [+ max pool: the un-pooled activation is never written]; the pre-BN bias only moves the mean, so it is folded into
    the moving average instead of being added); backward = pn2_bn_relu_backward (pool / ReLU masks, dgamma, dbeta, dy)
    -> dX = dY @ w^T, dW on pn2_linear_wgrad.  Replaces nine elementwise / reduction kernels per layer."""

    @staticmethod
    def forward(ctx, x2d, w, b, gamma, beta, running_mean, running_var, decay, relu, pool, defer, front=None):
        c = w.shape[1]
        pooled = pool > 1
        # producer of this layer's input (if it was an un-pooled dense+BN layer of this forward pass)
        prev = _bn_links.get(x2d.data_ptr()) if (USE_DGRAD_BN_STATS and x2d is not None) else None
        prev = prev if (prev is not None and prev.y is not None and prev.shape == tuple(x2d.shape)) else None
        xf = prev is not None and prev.sc is not None  # x2d is the producer's UN-normalised output
        defer = defer and not pooled and any(ctx.needs_input_grad)  # no tape node, nobody to keep the record: normalise here
        #                                    (the callers also drop the request when no tape is being recorded at all)
        if xf and c % 32 != 0:
            raise RuntimeError("a deferred batch-norm output reached a layer that cannot apply it")
        consts, folded = None, False
        if front is not None:
            # the first layer of an SA module with few point channels: gather + centre + concat + product + statistics + their
            # fold / constants in ONE launch (pn2_sa_first_layer_bn); x2d = the grouped input it leaves for the weight gradient
            xyz, new_xyz, points, idx = front
            bsz, n = xyz.shape[0], xyz.shape[1]
            m, ns = idx.shape[1], idx.shape[2]
            cpts = points.shape[2]
            ws = _bn_zeroed_scratch(c, xyz.device)
            y = torch.empty((bsz * m * ns, c), dtype=torch.float32, device=xyz.device)
            x2d = torch.empty((bsz * m * ns, 3 + cpts), dtype=torch.float32, device=xyz.device)
            if defer:
                save_mean = torch.empty(c, dtype=torch.float32, device=xyz.device)
                consts = (save_mean, torch.empty_like(save_mean), torch.empty_like(save_mean), torch.empty_like(save_mean))
            cs = consts if consts is not None else (None, None, None, None)
            with torch.cuda.device(xyz.device):
                check(lib.pn2_sa_first_layer_bn(bsz, n, m, ns, cpts, c, ptr(xyz), ptr(new_xyz), ptr(points), ptr(idx), ptr(w.contiguous()),
                                                ptr(y), ptr(x2d), ptr(ws), ws.numel() * ws.element_size(), 2 if defer else 1,
                                                ptr(gamma), ptr(beta), ptr(b), BN_EPSILON, float(decay), ptr(running_mean),
                                                ptr(running_var), ptr(cs[0]), ptr(cs[1]), ptr(cs[2]), ptr(cs[3]), stream_ptr()),
                      "pn2_sa_first_layer_bn")
            folded = True
        elif (xf or (USE_GEMM_BN_STATS and c % 32 == 0)) and USE_BN_FINISH_IN_PRODUCER:
            # ONE launch: GEMM (batch norm of the layer below applied on load when it was deferred) + column sums of y + -- in the
            # launch's last workgroup -- their fold and, for a layer that defers its own batch norm, its constants
            ws = _bn_zeroed_scratch(c, x2d.device)
            y, consts = hip_matmul_bn_stats_fin(x2d, w, ws, (prev.sc, prev.sh, prev.relu) if xf else None, 2 if defer else 1,
                                                gamma, beta, b, decay, running_mean, running_var)
            folded = True
        elif xf:
            ws = _bn_zeroed_scratch(c, x2d.device)
            y = hip_matmul_bn_stats_xf(x2d, w, ws, prev.sc, prev.sh, prev.relu)
        elif USE_GEMM_BN_STATS and c % 32 == 0:
            # the GEMM's epilogue leaves the column sums of y in the batch-norm workspace: no statistics pass over y
            ws = _bn_zeroed_scratch(c, x2d.device)
            y = hip_matmul_bn_stats(x2d, w, ws)
        else:
            ws, y = None, hip_matmul(x2d, w)
        if xf:
            prev.consumed = True
        ctx.xf = (prev.sc, prev.sh, bool(prev.relu)) if xf else None
        ctx.relu, ctx.pool = bool(relu), int(pool)
        ctx.prev = prev if (prev is not None and w.shape[1] > 16) else None
        gx = bool(USE_BN_GRAD_ON_LOAD and c % 4 == 0 and 16 < c <= 512 and not pooled)  # what backward will do (see _gx_usable)
        if defer:
            if consts is not None:
                save_mean, save_invstd, sc, sh = consts
            else:
                save_mean, save_invstd, sc, sh = _bn_train_forward_deferred(y, b, gamma, beta, running_mean, running_var, decay, ws)
            ctx.save_for_backward(x2d, w, y, gamma, beta, save_mean, save_invstd)
            ctx.link = _bn_register_producer(y, y, gamma, beta, save_mean, save_invstd, relu, False, sc, sh, gx=gx)
            if ctx.link is None:
                raise RuntimeError("deferred batch norm needs the producer links (USE_DGRAD_BN_STATS)")
            return y  # un-normalised: only the next dense layer of the stack may consume it
        z, ties, save_mean, save_invstd = _bn_train_forward(y, b, gamma, beta, running_mean, running_var, decay, relu, pool, ws,
                                                            folded=folded)
        if pooled:
            ctx.save_for_backward(x2d, w, y, gamma, beta, save_mean, save_invstd, z, ties)
        else:
            ctx.save_for_backward(x2d, w, y, gamma, beta, save_mean, save_invstd)
        ctx.link = _bn_register_producer(z, y, gamma, beta, save_mean, save_invstd, relu, pooled, gx=gx)
        return z

    @staticmethod
    def backward(ctx, dz):
        x2d, w, y, gamma, beta, save_mean, save_invstd = ctx.saved_tensors[:7]
        zmax, ties = ctx.saved_tensors[7:] if ctx.pool > 1 else (None, None)
        dz = dz.contiguous()
        if _gx_usable(y.shape[0], y.shape[1], ctx.pool, dz, y):
            # the gradient leaving the batch norm is formed by the two GEMMs while they load (y, dz): never written
            coef, dgamma, dbeta = _bn_grad_constants(dz, y, gamma, beta, save_mean, save_invstd, ctx.relu, ctx.pool, zmax, ties,
                                                     ctx.link)
            dx = None
            pv = ctx.prev if (ctx.prev is not None and ctx.prev.y is not None) else None
            if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
                both = _hip_bwd_fused(x2d, ctx.xf, y, dz, coef, ctx.relu, ctx.pool, zmax, ties, w, pv)  # narrow layers: one launch
                if both is not None:
                    return both[0], both[1], None, dgamma, dbeta, None, None, None, None, None, None, None
            if ctx.needs_input_grad[0]:
                dx = _hip_dgrad_gx(y, dz, coef, ctx.relu, ctx.pool, zmax, ties, w, pv)
            dw = _hip_wgrad_gx(x2d, ctx.xf, y, dz, coef, ctx.relu, ctx.pool, zmax, ties, w) if ctx.needs_input_grad[1] else None
            return dx, dw, None, dgamma, dbeta, None, None, None, None, None, None, None
        dy, dgamma, dbeta = _bn_train_backward(dz, y, gamma, beta, save_mean, save_invstd, ctx.relu, ctx.pool, zmax, ties, ctx.link)
        dx = None
        if ctx.needs_input_grad[0]:
            pv = ctx.prev
            dx = hip_linear_dgrad_linked(dy, w, pv) if (pv is not None and pv.y is not None) else hip_linear_dgrad(dy, w)
        dw = None
        if ctx.needs_input_grad[1]:
            dw = _hip_wgrad(x2d, dy, w, ctx.xf)
        # a constant in front of batch norm has no effect on the output: its gradient is exactly zero -- None, which the
        # trainer's gradient buffer treats as (and keeps at) zero without a fill per layer
        return dx, dw, None, dgamma, dbeta, None, None, None, None, None, None, None


USE_SA_FIRST_LAYER_FUSED = True  # SA module, few point channels: front end + first conv + statistics in one launch (A/B, tests)


def conv2d_sa_first_small(xyz, new_xyz, points, idx, num_output_channels, scope, bn_decay=None, pool=0, defer_bn=False):
    """conv2d(sample_and_group's [grouped_xyz - new_xyz | grouped points], ..., bn=True, is_training=True, relu) -- the FIRST layer
    of an SA module whose points carry at most 5 channels and no gradient (the level-0 module's colours) -- without building the
    grouped tensor first: _TrainDenseBnRelu(front=...) on pn2_sa_first_layer_bn.  Same variables as tf_util.conv2d under `scope`.
    -> (b, m, nsample or 1, cout)"""
    cout = int(num_output_channels)
    cin = 3 + points.shape[2]
    with variable_scope(scope):
        st, w, b, bnv = _dense_variables(cin, cout, True, (1, 1, cin, cout))
        st.train_epoch += 1
        beta, gamma, mean, var = bnv
        decay = 0.9 if bn_decay is None else float(bn_decay)
        pool = int(pool) if pool and pool > 1 else 0
        front = (xyz.detach().contiguous(), new_xyz.detach().contiguous(), points.detach().contiguous(), idx.contiguous())
        z = _TrainDenseBnRelu.apply(None, w.reshape(cin, cout), b, gamma, beta, mean, var, decay, True, pool,
                                    bool(defer_bn) and not pool and torch.is_grad_enabled(), front)
    m, ns = idx.shape[1], idx.shape[2]
    return z.reshape(xyz.shape[0], m, ns // pool if pool else ns, cout)


HOIST_BN_STATS_MIN_ROWS = 32768
USE_HOIST_BN_STATS = True  # ... and its batch statistics (+ fold + constants) taken by the launch that writes it (A/B, tests)
USE_HOISTED_TRAIN = True  # first layer of SA2-SA4 / FP4 with its feature half applied to the source rows (A/B, tests)
# (FP1-FP3, whose skip link is a wide SA feature tensor that needs a GEMM of its own, keep the concatenated form: the hoisted
# variant was built and measured slower -- 4.13 vs 4.05 ms per step, DESIGN.md section 9 -- and removed in round 4.)


def _hip_wgrad_into(x2d, dy, dw_rows):
    """dw_rows (a zero-filled row block of a weight-gradient tile) += x2d^T @ dy on pn2_linear_wgrad_accumulate"""
    with torch.cuda.device(dy.device):
        check(lib.pn2_linear_wgrad_accumulate(x2d.shape[0], x2d.shape[1], dy.shape[1], ptr(x2d), ptr(dy), ptr(dw_rows), stream_ptr()),
              "pn2_linear_wgrad_accumulate")


class _TrainHoistedBnRelu(torch.autograd.Function):
    """First layer of an SA / FP module, training path, with the feature half of the 1x1 conv applied to the SOURCE rows
    (gather / interpolation are linear and commute with it; csrc/pn2_hoist.hip):
        SA: y = (group_point(xyz, idx) - new_xyz) @ W[:3] + (points @ W[3:])[idx]              (pointnet_util.py:39-54,150-156)
        FP: y = three_interpolate(points2 @ W[:c2], idx, w(dist)) + points1 @ W[c2:]             (pointnet_util.py:300-312)
    then batch norm + ReLU (+ max pool) as in _TrainDenseBnRelu.  The grouped / concatenated tensor is never built, and the
    GEMM, its data gradient and its weight gradient run on n (resp. m) rows instead of m * nsample (resp. n).  Backward:
    dy from the batch-norm kernels; dz = the scatter of dy through the precomputed plan (the same list the gradient of
    group_point / three_interpolate uses); d(points) = dz @ Wb^T, dWb = points^T dz on the source rows; dWa = a^T dy with
    a = the centred coordinates (SA) resp. points1 (FP: at most 8 skip-link channels that carry no gradient -- the colours
    of the level-0 module)."""

    @staticmethod
    def forward(ctx, src, w, b, gamma, beta, running_mean, running_var, decay, relu, pool, kind, g0, g1, g2, plan, defer):
        bsz, nsrc, c = src.shape
        cout = w.shape[1]
        src2d = src.reshape(-1, c)
        if kind == "sa":
            xyz, new_xyz, idx = g0, g1, g2
            m, ns = idx.shape[1], idx.shape[2]
            wa, wb = w[:3], w[3:]
            rows_b = m * ns
        else:
            dist, idx, points1 = g0, g1, g2
            n, c1 = points1.shape[1], points1.shape[2]
            wb, wa = w[:c], w[c:]
            rows_b = n
        z = hip_matmul(src2d, wb)  # (b * nsrc, cout)
        y = torch.empty((bsz * rows_b, cout), dtype=torch.float32, device=src.device)
        pooled = pool > 1
        defer = defer and any(ctx.needs_input_grad)  # as _TrainDenseBnRelu
        # the deferred batch norm's statistics (+ their fold and the constants) ride in the launch that writes y
        # (from 32768 rows on: below, the epilogue's tail -- 2 * cout fp64 atomics per workgroup, the ticket, the fold of 64 slot copies
        #  by one workgroup -- costs more than the statistics pass it replaces: 8192 x 256: 19 vs 15 us, tools/hoist_stats_ab.py)
        fused_stats = bool(defer and not pooled and USE_HOIST_BN_STATS and USE_BN_FINISH_IN_PRODUCER and bsz * rows_b >= HOIST_BN_STATS_MIN_ROWS)
        bn_args = ()
        if fused_stats:
            ws = _bn_zeroed_scratch(cout, src.device)
            save_mean = torch.empty(cout, dtype=torch.float32, device=src.device)
            save_invstd, sc, sh = torch.empty_like(save_mean), torch.empty_like(save_mean), torch.empty_like(save_mean)
            bn_args = (ptr(ws), ws.numel() * ws.element_size(), 2, ptr(gamma), ptr(beta), ptr(b), BN_EPSILON, float(decay),
                       ptr(running_mean), ptr(running_var), ptr(save_mean), ptr(save_invstd), ptr(sc), ptr(sh))
        with torch.cuda.device(src.device):
            if kind == "sa":
                a = torch.empty((bsz * rows_b, 3), dtype=torch.float32, device=src.device)
                fn, nm = (lib.pn2_sa_hoist_rows_bn, "pn2_sa_hoist_rows_bn") if fused_stats else (lib.pn2_sa_hoist_rows, "pn2_sa_hoist_rows")
                check(fn(bsz, nsrc, m, ns, cout, ptr(xyz), ptr(new_xyz), ptr(idx), ptr(z), ptr(wa), ptr(y), ptr(a), *bn_args,
                         stream_ptr()), nm)
            else:
                a = points1.reshape(-1, c1)
                if c1 > 8:
                    raise ValueError("the hoisted FP front end takes at most 8 skip-link channels")
                fn, nm = (lib.pn2_fp_hoist_rows_bn, "pn2_fp_hoist_rows_bn") if fused_stats else (lib.pn2_fp_hoist_rows, "pn2_fp_hoist_rows")
                check(fn(bsz, n, nsrc, c1, cout, ptr(dist), ptr(idx), ptr(a), ptr(z), ptr(wa), ptr(y), *bn_args, stream_ptr()), nm)
        ctx.relu, ctx.pool, ctx.kind, ctx.dims = bool(relu), int(pool), kind, (bsz, nsrc, rows_b, c)
        if defer and not pooled:
            if not fused_stats:
                save_mean, save_invstd, sc, sh = _bn_train_forward_deferred(y, b, gamma, beta, running_mean, running_var, decay)
            ctx.save_for_backward(src2d, w, y, gamma, beta, save_mean, save_invstd, a, plan)
            ctx.link = _bn_register_producer(y, y, gamma, beta, save_mean, save_invstd, relu, False, sc, sh)
            if ctx.link is None:
                raise RuntimeError("deferred batch norm needs the producer links (USE_DGRAD_BN_STATS)")
            return y
        zact, ties, save_mean, save_invstd = _bn_train_forward(y, b, gamma, beta, running_mean, running_var, decay, relu, pool)
        ctx.save_for_backward(src2d, w, y, gamma, beta, save_mean, save_invstd, a, plan, *((zact, ties) if pooled else ()))
        ctx.link = _bn_register_producer(zact, y, gamma, beta, save_mean, save_invstd, relu, pooled)
        return zact

    @staticmethod
    def backward(ctx, dz):
        src2d, w, y, gamma, beta, save_mean, save_invstd, a, plan = ctx.saved_tensors[:9]
        zmax, ties = ctx.saved_tensors[9:] if ctx.pool > 1 else (None, None)
        bsz, nsrc, rows_b, c = ctx.dims
        cout = w.shape[1]
        dy, dgamma, dbeta = _bn_train_backward(dz, y, gamma, beta, save_mean, save_invstd, ctx.relu, ctx.pool, zmax, ties, ctx.link)
        sa = ctx.kind == "sa"
        wa, wb = (w[:3], w[3:]) if sa else (w[c:], w[:c])
        from .pointnet_util import _scatter_plan_apply
        dzs = _scatter_plan_apply(plan, dy.view(bsz, rows_b, cout), 0, cout, rows_b if sa else 3 * rows_b, 1 if sa else 3, nsrc)
        dzs2d = dzs.view(-1, cout)
        dsrc = hip_linear_dgrad(dzs2d, wb).view(bsz, nsrc, c) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            dw = get_default_store().grad_view(w)
            if dw is None:
                arena = get_default_store().zero_arena
                v = arena.take(w.numel() * 4) if arena is not None else None
                dw = v[:w.numel() * 4].view(torch.float32).view_as(w) if v is not None else torch.zeros_like(w)
            dwa, dwb = (dw[:3], dw[3:]) if sa else (dw[c:], dw[:c])
            _hip_wgrad_into(src2d, dzs2d, dwb)
            _hip_wgrad_into(a, dy, dwa)
        return (dsrc, dw, None, dgamma, dbeta) + (None,) * 11


def conv2d_hoisted_first(kind, src, geo, plan, cin, num_output_channels, scope, bn_decay=None, pool=0, defer_bn=False):
    """conv2d(concat-of-a-gathered-tensor, ..., bn=True, is_training=True, activation relu) -- the FIRST layer of an SA
    (kind "sa": geo = (xyz, new_xyz, idx)) or FP (kind "fp": geo = (dist, idx, points1)) module -- without building the
    gathered tensor: _TrainHoistedBnRelu.  Same variables (names, shapes, initialisation) as tf_util.conv2d under `scope`;
    cin = width of the tensor the reference convolves (3 + c resp. c2 + c1).
    -> (b, m, nsample or nsample/pool, cout) resp. (b, n, 1, cout)."""
    cout = int(num_output_channels)
    with variable_scope(scope):
        st, w, b, bnv = _dense_variables(cin, cout, True, (1, 1, cin, cout))
        st.train_epoch += 1
        beta, gamma, mean, var = bnv
        decay = 0.9 if bn_decay is None else float(bn_decay)
        pool = int(pool) if pool and pool > 1 else 0
        z = _TrainHoistedBnRelu.apply(src.contiguous(), w.reshape(cin, cout), b, gamma, beta, mean, var, decay, True, pool, kind,
                                      geo[0].contiguous(), geo[1].contiguous(), geo[2].contiguous(), plan,
                                      bool(defer_bn) and not pool and torch.is_grad_enabled())
    if kind == "sa":
        m, ns = geo[2].shape[1], geo[2].shape[2]
        return z.reshape(src.shape[0], m, ns // pool if pool else ns, cout)
    return z.reshape(src.shape[0], geo[2].shape[1], 1, cout)


def _train_layer(inputs, w2d, b, bnv, bn_decay, relu, pool=0, defer=False):
    """One dense layer of the training path, entirely on the HIP library: inputs (..., cin) -> (..., cout); pool > 1 also
    takes the max over groups of `pool` consecutive entries of the second-to-last axis (..., W, cin) -> (..., W/pool, cout).
    With batch norm: _TrainDenseBnRelu; without: _TrainMatmul (activation None: the class head) or _TrainDenseRelu (bn=False
    layers of the layer API, tf_util.py:186-204).  There is no torch fallback: batch-norm widths beyond 1024, or not a multiple of
    4 above 256, run as independent column blocks on the same kernels.  (tests/torch_layers.py holds the plain-torch reference
    of this function.)"""
    cin, cout = w2d.shape
    pool = int(pool) if pool and pool > 1 else 0
    lead = list(inputs.shape[:-1])
    if pool:
        if lead[-1] % pool:
            raise ValueError("pool must divide the grouped axis")
        lead[-1] //= pool
    require_cuda(inputs)
    if inputs.dtype != torch.float32:
        raise TypeError("the training path is float32")
    if bnv is None:
        if pool:  # the fused max over K lives in the batch-norm kernels; a layer without batch norm is pooled by its caller
            raise ValueError("pool > 1 needs bn=True on the training path (pointnet_sa_module pools bn=False stacks with group_pool)")
        if _deferred_producer(inputs.reshape(-1, cin)) is not None:
            raise RuntimeError("a deferred batch-norm output reached a layer without batch norm")
        return _train_dense(inputs, w2d, b, relu=relu)
    if not (cout <= 1024 and (cout % 4 == 0 or cout <= 256)):
        # A width the batch-norm kernels do not take in one piece (> 1024 channels, or > 256 and not a multiple of 4; the reference's
        # batch_norm_template takes any width, tf_util.py:555-581).  Batch norm, ReLU and the max over K act per channel, so the
        # layer IS the concatenation of independent layers on column blocks of (W, gamma, beta, moving averages): blocks of
        # <= 1024 channels (multiples of 4) and one remainder of <= 256.  Slow -- one GEMM and one set of batch-norm launches per
        # block, a column copy in and a concatenation out -- but every value comes from the same HIP kernels.
        beta, gamma, mean, var = bnv
        cuts, c0 = [], 0
        while cout - c0 > 256 and (cout - c0) % 4 != 0 or cout - c0 > 1024:
            step = min(1024, (cout - c0) // 4 * 4)
            if (cout - c0 - step) > 256 and step < 1024:  # keep the remainder <= 256
                step = (cout - c0 - 256 + 3) // 4 * 4
            cuts.append((c0, c0 + step))
            c0 += step
        cuts.append((c0, cout))
        outs = [_train_layer(inputs, w2d[:, a:e].contiguous(), None if b is None else b[a:e],
                             (beta[a:e], gamma[a:e], mean[a:e], var[a:e]), bn_decay, relu, pool, defer=False) for a, e in cuts]
        return torch.cat(outs, dim=-1)
    beta, gamma, mean, var = bnv
    decay = 0.9 if bn_decay is None else float(bn_decay)  # tf_util.py:571
    z = _TrainDenseBnRelu.apply(inputs.reshape(-1, cin).contiguous(), w2d.contiguous(), b, gamma, beta, mean, var,
                                decay, relu, pool, bool(defer) and not pool and relu and torch.is_grad_enabled())
    return z.reshape(lead + [cout])


def conv2d(inputs, num_output_channels, kernel_size, scope, stride=(1, 1), padding="SAME", data_format="NHWC",
           use_xavier=True, stddev=1e-3, weight_decay=None, activation_fn=torch.relu, bn=False, bn_decay=None,
           is_training=None, pool=0, defer_bn=False):
    """1x1 conv over an NHWC tensor (B,H,W,C) = matmul over C (tf_util.py:128-204).
    Only the configuration the SA/FP stack uses is implemented: kernel [1,1],
    stride [1,1], NHWC.  `pool` (extension): max over groups of `pool` rows of W,
    fused in the HIP epilogue on the inference path."""
    if list(kernel_size) != [1, 1] or list(stride) != [1, 1] or data_format != "NHWC":
        raise NotImplementedError("only 1x1 / stride 1 / NHWC conv2d is on the SA/FP path")
    if activation_fn not in (torch.relu, None):
        raise NotImplementedError("activation_fn must be relu or None")
    cin = inputs.shape[-1]
    cout = int(num_output_channels)
    with variable_scope(scope):
        if not is_training:
            w2, b2 = folded_dense(cin, cout, bn, (1, 1, cin, cout), pad_to=32)
            y = hip_linear(inputs.reshape(-1, cin), w2, b2, relu=activation_fn is not None, pool=pool)
            if y.shape[1] != cout:
                y = y[:, :cout]
            lead = list(inputs.shape[:-1])
            if pool and pool > 1:
                lead[-1] //= pool
            return y.reshape(lead + [cout])
        st, w, b, bnv = _dense_variables(cin, cout, bn, (1, 1, cin, cout))
        st.train_epoch += 1
        # defer_bn (extension, training): hand the UN-normalised output to the next conv2d of the stack (see USE_BN_ON_LOAD)
        return _train_layer(inputs, w.reshape(cin, cout), b, bnv, bn_decay, activation_fn is not None, pool,
                            defer=bool(defer_bn) and bn)


def conv1d(inputs, num_output_channels, kernel_size, scope, stride=1, padding="SAME", use_xavier=True,
           stddev=1e-3, weight_decay=None, activation_fn=torch.relu, bn=False, bn_decay=None, is_training=None):
    """kernel-1 conv over (B,N,C) (tf_util.py:54-125): same maths as conv2d on (B,N,1,C)."""
    if kernel_size != 1 or stride != 1:
        raise NotImplementedError("only kernel 1 / stride 1 conv1d is on the SA/FP path")
    cin = inputs.shape[-1]
    cout = int(num_output_channels)
    with variable_scope(scope):
        if not is_training:
            if cout <= 16 and not bn and activation_fn is None:  # the class head (model.py:145-146): one streaming launch
                st, w, b, _ = _dense_variables(cin, cout, False, (1, cin, cout))
                y = hip_linear_narrow(inputs.reshape(-1, cin), w.detach().reshape(cin, cout), b.detach())
                if y is not None:
                    return y.reshape(list(inputs.shape[:-1]) + [cout])
            w2, b2 = folded_dense(cin, cout, bn, (1, cin, cout), pad_to=32)
            y = hip_linear(inputs.reshape(-1, cin), w2, b2, relu=activation_fn is not None)
            if y.shape[1] != cout:
                y = y[:, :cout].contiguous()
            return y.reshape(list(inputs.shape[:-1]) + [cout])
        st, w, b, bnv = _dense_variables(cin, cout, bn, (1, cin, cout))
        st.train_epoch += 1
        y = _train_layer(inputs, w.reshape(cin, cout), b, bnv, bn_decay, activation_fn is not None)
        return y


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, keep_prob, state):
        x = x.contiguous()
        y = torch.empty_like(x)
        mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            check(lib.pn2_dropout(x.numel(), ptr(x), float(keep_prob), ptr(state), ptr(y), ptr(mask), stream_ptr()),
                  "pn2_dropout")
        ctx.save_for_backward(mask)
        ctx.keep = float(keep_prob)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        with torch.cuda.device(dy.device):
            check(lib.pn2_dropout_grad(dy.numel(), ptr(dy), ptr(mask), ctx.keep, ptr(dx), stream_ptr()), "pn2_dropout_grad")
        return dx, None, None


def dropout(inputs, is_training, scope, keep_prob=0.5, noise_shape=None):
    """tf_util.py:646-665: active only while training.  The draw is a pure function of (store seed ^ scope, store
    step counter, element index) read from device memory (pn2_dropout): replayable inside a captured hipGraph; the
    trainer advances `store.rng_state[1]` once per step."""
    if not is_training:
        return inputs
    if noise_shape is not None:
        raise NotImplementedError("noise_shape is not used on the SA/FP path")
    require_cuda(inputs)
    st = get_default_store()
    import zlib
    key = _full_name(scope)
    state = st.dropout_state(key, zlib.crc32(key.encode()))
    assert_not_deferred(inputs, "dropout")
    return _Dropout.apply(inputs, keep_prob, state)
