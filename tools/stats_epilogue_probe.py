"""What the batch-statistics epilogue of the forward GEMM costs (pn2_linear_bn_stats_fin against pn2_linear at the same shape), run once
per experimental build of the library (PN2_HIP_LIBRARY): the shipped one, fp32 partial sums per tile, no arithmetic, no atomics.
    gpurun -- 'for v in hip f32 nomath noatomic; do PN2_HIP_LIBRARY=$PWD/open3d-pointnet2-semantic3d_amd/libpn2_$v.so python tools/stats_epilogue_probe.py; done'"""
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


out = [os.path.basename(os.environ.get("PN2_HIP_LIBRARY", "libpn2_hip.so"))]
for rows, cin, cout in [(131072, 128, 128), (131072, 64, 128), (524288, 32, 64), (32768, 128, 256)]:
    x = torch.randn(rows, cin, device=dev); w = torch.randn(cin, cout, device=dev) / cin ** 0.5
    y = torch.empty(rows, cout, device=dev)
    sc, sh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1
    g, b = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
    sm, si, s2, h2 = (torch.empty(cout, device=dev) for _ in range(4))
    nb = L.pn2_bn_workspace_bytes(cout)
    ws = torch.zeros(nb // 8, dtype=torch.float64, device=dev)
    plain = lambda st: L.pn2_linear(rows, cin, cout, P(x), P(w), None, 0, 0, P(y), st)  # noqa: E731
    def fin(st):
        ws.zero_()
        return L.pn2_linear_bn_stats_fin(rows, cin, cout, P(x), P(w), P(y), P(ws), nb, P(sc), P(sh), 1, 2, P(g), P(b), None, cf(1e-3), cf(0.9),
                                         P(rm), P(rv), P(sm), P(si), P(s2), P(h2), st)
    def fin_noxf(st):
        ws.zero_()
        return L.pn2_linear_bn_stats_fin(rows, cin, cout, P(x), P(w), P(y), P(ws), nb, None, None, 0, 2, P(g), P(b), None, cf(1e-3), cf(0.9),
                                         P(rm), P(rv), P(sm), P(si), P(s2), P(h2), st)
    out.append("(%d,%d,%d) plain %.1f  stats+fin %.1f  xf+stats+fin %.1f" % (rows, cin, cout, graph_time(plain), graph_time(fin_noxf) - 2.9, graph_time(fin) - 2.9))
print("   ".join(out))
