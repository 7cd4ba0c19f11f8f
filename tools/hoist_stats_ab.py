"""pn2_sa_hoist_rows / pn2_fp_hoist_rows against their *_bn forms (statistics + fold + constants in the same launch) and against the
two launches they replace (+ pn2_bn_relu_forward_deferred), at the hoisted first layers of configs[1]'s training step.  Graph-timed.
    gpurun -- 'python tools/hoist_stats_ab.py'"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pn2_amd as pn2  # noqa: E402
from bench import s_scene  # noqa: E402
L, P = pn2._lib._raw, pn2._lib.ptr
dev = torch.device("cuda:0")
cf = ctypes.c_float


def graph_time(fn, reps=10, iters=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        st = ctypes.c_void_p(s.cuda_stream)
        fn(st); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn(st)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(iters):
            e0.record(s); g.replay(); e1.record(s); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


for kind, b, n, m, ns, c1, cout in [("sa", 16, 1024, 256, 32, 0, 64), ("sa", 16, 256, 64, 32, 0, 128), ("sa", 16, 64, 16, 32, 0, 256),
                                    ("fp", 16, 8192, 1024, 0, 3, 128)]:
    rs = np.random.RandomState(1)
    if kind == "sa":
        xyz = torch.from_numpy(s_scene(3, b, n)[..., :3].copy()).to(dev)
        new_xyz, idx = pn2.util.pointnet_util.sa_geometry(xyz, m, 0.9 * (1024 / n) ** 0.5, ns)
        z = torch.randn(b, n, cout, device=dev); wa = torch.randn(3, cout, device=dev)
        rows = b * m * ns
        a = torch.empty(rows, 3, device=dev)
        plain = lambda y, st: L.pn2_sa_hoist_rows(b, n, m, ns, cout, P(xyz), P(new_xyz), P(idx), P(z), P(wa), P(y), P(a), st)  # noqa: E731
        fused = lambda y, st, *bn: L.pn2_sa_hoist_rows_bn(b, n, m, ns, cout, P(xyz), P(new_xyz), P(idx), P(z), P(wa), P(y), P(a), *bn, st)  # noqa: E731
    else:
        xyz = torch.from_numpy(s_scene(4, b, n)[..., :3].copy()).to(dev)
        dist, idx = pn2.three_nn(xyz, xyz[:, :m].contiguous())
        z = torch.randn(b, m, cout, device=dev); p1 = torch.rand(b, n, c1, device=dev); wa = torch.randn(c1, cout, device=dev)
        rows = b * n
        plain = lambda y, st: L.pn2_fp_hoist_rows(b, n, m, c1, cout, P(dist), P(idx), P(p1), P(z), P(wa), P(y), st)  # noqa: E731
        fused = lambda y, st, *bn: L.pn2_fp_hoist_rows_bn(b, n, m, c1, cout, P(dist), P(idx), P(p1), P(z), P(wa), P(y), *bn, st)  # noqa: E731
    y = torch.empty(rows, cout, device=dev)
    g_, b_ = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
    sm, si, sc, sh = (torch.empty(cout, device=dev) for _ in range(4))
    nb = L.pn2_bn_workspace_bytes(cout)
    ws = torch.zeros(nb // 8, dtype=torch.float64, device=dev)
    def two(st):
        plain(y, st)
        return L.pn2_bn_relu_forward_deferred(rows, cout, P(y), P(g_), P(b_), None, cf(1e-3), cf(0.5), 0, P(rm), P(rv), P(ws), nb, P(sm),
                                              P(si), P(sc), P(sh), st)
    def one(st):
        ws.zero_()
        return fused(y, st, P(ws), nb, 2, P(g_), P(b_), None, cf(1e-3), cf(0.5), P(rm), P(rv), P(sm), P(si), P(sc), P(sh))
    def zero_only(st):
        ws.zero_()
    t_plain = graph_time(lambda st: plain(y, st))
    t_two, t_one, t_zero = graph_time(two), graph_time(one), graph_time(zero_only)
    print("%s b%d n%d m%d ns%d cout%d rows %d:  hoist %.1f us | hoist + deferred statistics %.1f | one launch %.1f (of which the memset %.1f)"
          % (kind, b, n, m, ns, cout, rows, t_plain, t_two, t_one, t_zero))
