"""The streaming forward kernel of the narrow training layers (csrc/pn2_fwd_narrow.h) against linear_kernel, same entry points
(pn2_linear_bn_stats_fin without / with the load transform), tuning-build hook 17 off / on; outputs bit for bit, statistics to fp64
summation order.  Graph-timed.
    python tools/dbg/build_both.py && gpurun -- 'PN2_HIP_LIBRARY=$PWD/open3d-pointnet2-semantic3d_amd/libpn2_tune.so python tools/fwd_narrow_ab.py'"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pn2_amd as pn2  # noqa: E402
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


for rows, cin, cout in [(524288, 32, 32), (524288, 32, 64), (131072, 64, 64), (131072, 64, 128), (131072, 32, 128), (65536, 64, 32)]:
    torch.manual_seed(1)
    x = torch.randn(rows, cin, device=dev); w = torch.randn(cin, cout, device=dev) / cin ** 0.5
    sc, sh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1
    g, b = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    nb = L.pn2_bn_workspace_bytes(cout)
    res = {}
    for on in (0, 1):
        L.pn2_debug_set(17, on)
        outs, times = [], []
        for xfm in (False, True):
            y = torch.empty(rows, cout, device=dev)
            ws = torch.zeros(nb // 8, dtype=torch.float64, device=dev)
            rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
            sm, si, s2, h2 = (torch.empty(cout, device=dev) for _ in range(4))
            def fin(st):
                ws.zero_()
                return L.pn2_linear_bn_stats_fin(rows, cin, cout, P(x), P(w), P(y), P(ws), nb, P(sc) if xfm else None, P(sh) if xfm else None,
                                                 1, 2, P(g), P(b), None, cf(1e-3), cf(0.9), P(rm), P(rv), P(sm), P(si), P(s2), P(h2), st)
            assert fin(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
            torch.cuda.synchronize()
            outs += [y.clone(), sm.clone(), si.clone()]
            times.append(graph_time(fin) - 2.9)
        res[on] = (outs, times)
    eq = [bool(torch.equal(a, b_)) for a, b_ in zip(res[0][0], res[1][0])]
    rel = [float(((a - b_).abs() / (b_.abs() + 1e-6)).max()) for a, b_ in zip(res[0][0], res[1][0])]
    mb = rows * (cin + cout) * 4 / 1e6
    print("(%d, %d -> %d)  stats+fin: linear_kernel %.1f us -> streaming %.1f us (%.2f TB/s)   xf+stats+fin: %.1f -> %.1f (%.2f TB/s)"
          % (rows, cin, cout, res[0][1][0], res[1][1][0], mb / res[1][1][0] * 1e-6 * 1e6 / 1e6, res[0][1][1], res[1][1][1], mb / res[1][1][1] * 1e-6 * 1e6 / 1e6))
    print("     y / mean / invstd equal [plain: %s %s %s] [xf: %s %s %s]  max rel diff of the moments %.1e" % (*eq, max(rel[1], rel[2], rel[4], rel[5])))
L.pn2_debug_set(17, 1)
