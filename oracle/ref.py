"""ctypes front-end of oracle/_ref: the REFERENCE's own tf_sampling.cu / tf_grouping.cu kernels built
for gfx950 (oracle/Makefile target `_ref`, doors in oracle/ref_shim.hip).

TEST INFRASTRUCTURE ONLY (tests/, smoke).  Needs a GPU: numpy in, numpy out, device buffers through
torch.  Argument orders follow the reference's Python op wrappers (tf_ops/tf_sampling.py:27,38,61;
tf_ops/tf_grouping.py:13,31,46).

Three builds:
    "off"        -ffp-contract=off                       == oracle arithmetic mode 0
    "fast_noslp" -ffp-contract=fast -fno-slp-vectorize   == oracle mode 1  fma(dz,dz,fma(dx,dx,dy*dy)) in ball query,
                                                            mode 2  fma(dz,dz,fma(dy,dy,dx*dx)) in FPS
                                                            (the LLVM DAG-combine contraction order; THE PRODUCT'S DEFAULTS)
    "fast"       -ffp-contract=fast (hipcc's default)    == oracle mode 5  fma(dy,dy,dx*dx)+dz*dz
                                                            (amdgpu SLP: packed squares, one fused)
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_HERE, "_ref")
BUILDS = ("off", "fast_noslp", "fast")
# oracle arithmetic mode each build must reproduce bit-for-bit (tests/test_ref_gpu.py proves it)
ORACLE_MODE = {"off": 0, "fast_noslp": 1, "fast": 5}
_libs = {}


def path(build):
    return os.path.join(_DIR, "libpn2_ref_%s.so" % build)


def available(build="off"):
    return os.path.exists(path(build))


def lib(build="off"):
    if build not in _libs:
        import torch  # noqa: F401  (maps libamdhip64 first so the runtime is shared)
        if not available(build):
            raise RuntimeError("oracle/_ref not built: run `make -C oracle _ref` where /root/reference exists")
        _libs[build] = ctypes.CDLL(path(build))
    return _libs[build]


def _dev(a, dtype):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError("oracle/_ref %s failed: hipError %d" % (what, rc))


def farthest_point_sample(npoint, inp, build="off"):
    import torch
    x = _dev(inp, np.float32)
    b, n, _ = x.shape
    temp = torch.empty((32, n), dtype=torch.float32, device="cuda")  # tf_sampling.cpp:146-148
    out = torch.empty((b, npoint), dtype=torch.int32, device="cuda")
    _chk(lib(build).ref_farthest_point_sample(b, n, int(npoint), _p(x), _p(temp), _p(out)), "fps")
    return out.cpu().numpy()


def gather_point(inp, idx, build="off"):
    import torch
    x, i = _dev(inp, np.float32), _dev(idx, np.int32)
    b, n, _ = x.shape
    m = i.shape[1]
    out = torch.empty((b, m, 3), dtype=torch.float32, device="cuda")
    _chk(lib(build).ref_gather_point(b, n, m, _p(x), _p(i), _p(out)), "gather_point")
    return out.cpu().numpy()


def gather_point_grad(inp, idx, out_g, build="off"):
    import torch
    i, g = _dev(idx, np.int32), _dev(out_g, np.float32)
    b, n, _ = np.shape(inp)
    m = i.shape[1]
    inp_g = torch.empty((b, n, 3), dtype=torch.float32, device="cuda")
    _chk(lib(build).ref_gather_point_grad(b, n, m, _p(g), _p(i), _p(inp_g)), "gather_point_grad")
    return inp_g.cpu().numpy()


def prob_sample(inp_p, inp_r, build="off"):
    import torch
    p, r = _dev(inp_p, np.float32), _dev(inp_r, np.float32)
    b, n = p.shape
    m = r.shape[1]
    temp = torch.empty((b, n), dtype=torch.float32, device="cuda")
    out = torch.empty((b, m), dtype=torch.int32, device="cuda")
    _chk(lib(build).ref_prob_sample(b, n, m, _p(p), _p(r), _p(temp), _p(out)), "prob_sample")
    return out.cpu().numpy(), temp.cpu().numpy()


def query_ball_point(radius, nsample, xyz1, xyz2, build="off"):
    import torch
    x1, x2 = _dev(xyz1, np.float32), _dev(xyz2, np.float32)
    b, n, _ = x1.shape
    m = x2.shape[1]
    idx = torch.zeros((b, m, nsample), dtype=torch.int32, device="cuda")  # empty balls: row left untouched
    cnt = torch.empty((b, m), dtype=torch.int32, device="cuda")
    _chk(lib(build).ref_query_ball_point(b, n, m, ctypes.c_float(radius), int(nsample), _p(x1), _p(x2),
                                         _p(idx), _p(cnt)), "query_ball_point")
    return idx.cpu().numpy(), cnt.cpu().numpy()


def selection_sort(k, dist, build="off"):
    import torch
    d = _dev(dist, np.float32)
    b, m, n = d.shape
    outi = torch.empty((b, m, n), dtype=torch.int32, device="cuda")
    out = torch.empty((b, m, n), dtype=torch.float32, device="cuda")
    _chk(lib(build).ref_selection_sort(b, n, m, int(k), _p(d), _p(outi), _p(out)), "selection_sort")
    return outi.cpu().numpy(), out.cpu().numpy()


def group_point(points, idx, build="off"):
    import torch
    p, i = _dev(points, np.float32), _dev(idx, np.int32)
    b, n, c = p.shape
    _, m, ns = i.shape
    out = torch.empty((b, m, ns, c), dtype=torch.float32, device="cuda")
    _chk(lib(build).ref_group_point(b, n, c, m, ns, _p(p), _p(i), _p(out)), "group_point")
    return out.cpu().numpy()


def group_point_grad(points, idx, grad_out, build="off"):
    import torch
    i, g = _dev(idx, np.int32), _dev(grad_out, np.float32)
    b, n, c = np.shape(points)
    _, m, ns = i.shape
    gp = torch.empty((b, n, c), dtype=torch.float32, device="cuda")
    _chk(lib(build).ref_group_point_grad(b, n, c, m, ns, _p(g), _p(i), _p(gp)), "group_point_grad")
    return gp.cpu().numpy()


# ---- the reference's two dependency-free HOST functions (tf_interpolate.cpp:307-330,397-421), lifted at build time by
#      oracle/lift_interpolate.py and compiled with the reference's host flags: oracle/_ref/libpn2_ref_interp.so.  CPU only.
_interp = None


def interp_available():
    return os.path.exists(os.path.join(_DIR, "libpn2_ref_interp.so"))


def _interp_lib():
    global _interp
    if _interp is None:
        if not interp_available():
            raise RuntimeError("oracle/_ref/libpn2_ref_interp.so not built: run `make -C oracle _ref` where /root/reference exists")
        _interp = ctypes.CDLL(os.path.join(_DIR, "libpn2_ref_interp.so"))
        _interp.threeinterpolate_cpu.restype = None
        _interp.threeinterpolate_grad_cpu.restype = None
    return _interp


def _np(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def three_interpolate(points, idx, weight):
    """threeinterpolate_cpu(b, m, c, n, points, idx, weight, out): points (b,m,c), idx / weight (b,n,3) -> (b,n,c)"""
    points, idx, weight = _np(points, np.float32), _np(idx, np.int32), _np(weight, np.float32)
    b, m, c = points.shape
    n = idx.shape[1]
    out = np.empty((b, n, c), np.float32)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    _interp_lib().threeinterpolate_cpu(b, m, c, n, vp(points), vp(idx), vp(weight), vp(out))
    return out


def three_interpolate_grad(points, idx, weight, grad_out):
    """ThreeInterpolateGradOp: memset(grad_points, 0) (tf_interpolate.cpp:477) + threeinterpolate_grad_cpu(b, n, c, m, ...)"""
    points, idx, weight = _np(points, np.float32), _np(idx, np.int32), _np(weight, np.float32)
    grad_out = _np(grad_out, np.float32)
    b, m, c = points.shape
    n = idx.shape[1]
    gp = np.zeros((b, m, c), np.float32)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    _interp_lib().threeinterpolate_grad_cpu(b, n, c, m, vp(grad_out), vp(idx), vp(weight), vp(gp))
    return gp
