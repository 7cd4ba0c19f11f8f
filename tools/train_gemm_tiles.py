"""Tile configurations of the three big training GEMM forms (tuning build: pn2_debug_set(8, cfg): 1 = 128 x 128 tile, 2 = 64 x 128,
3 = 32 x 128) at the 128-wide layers of the step: forward with the batch norm of the layer below applied on load + statistics,
the data gradient with the batch-norm gradient formed on load + the epilogue for the layer below, graph-timed.
    python open3d-pointnet2-semantic3d_amd/build.py --tuning && gpurun -- 'python tools/train_gemm_tiles.py'"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from fps_ab import timeit  # noqa: E402
import pn2_amd as pn2  # noqa: E402
raw, lib = pn2._lib._raw, pn2._lib.lib
tfu = pn2.util.tf_util
dev = torch.device("cuda:0")
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())  # noqa: E731
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for rows, cin, cout in [(131072, 128, 128), (131072, 64, 128), (32768, 128, 256), (524288, 32, 64)]:
    x = torch.randn(rows, cin, device=dev); w = torch.randn(cin, cout, device=dev) / cin ** 0.5
    sc, sh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1
    y = torch.empty(rows, cout, device=dev)
    g, b = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
    sm, si, s2, h2 = (torch.empty(cout, device=dev) for _ in range(4))
    nb = raw.pn2_bn_workspace_bytes(cout)
    ws = torch.zeros(nb // 8, dtype=torch.float64, device=dev)
    # data gradient: dx (rows, cin) = dy (rows, cout) @ w^T with dy on load, epilogue for the layer below (width cin)
    dz = torch.randn(rows, cout, device=dev); coef = torch.rand(6, cout, device=dev) + 0.5
    dx = torch.empty(rows, cin, device=dev)
    nb2 = raw.pn2_bn_workspace_bytes(cin)
    ws2 = torch.zeros(nb2 // 8, dtype=torch.float64, device=dev)
    gb, bb, mb, ib = torch.ones(cin, device=dev), torch.zeros(cin, device=dev), torch.zeros(cin, device=dev), torch.ones(cin, device=dev)
    cb, dgb, dbb = torch.empty(6, cin, device=dev), torch.empty(cin, device=dev), torch.empty(cin, device=dev)
    row = ["(%d,%d,%d)" % (rows, cin, cout)]
    for cfg in (0, 1, 2, 3):
        raw.pn2_debug_set(8, cfg)
        def fwd():
            ws.zero_()
            return raw.pn2_linear_bn_stats_fin(rows, cin, cout, P(x), P(w), P(y), P(ws), nb, P(sc), P(sh), 1, 2, P(g), P(b), None,
                                               ctypes.c_float(1e-3), ctypes.c_float(0.9), P(rm), P(rv), P(sm), P(si), P(s2), P(h2), st)
        def dgr():
            ws2.zero_()
            return raw.pn2_linear_dgrad_fin(rows, cin, cout, None, P(y), P(dz), P(coef), 1, 0, None, None, P(w), P(dx), P(x), P(gb), P(bb),
                                            P(mb), P(ib), 1, P(ws2), nb2, 3, P(cb), P(dgb), P(dbb), st)
        rc1, rc2 = fwd(), dgr()
        t1 = timeit(fwd, 20) if rc1 == 0 else float("nan")
        t2 = timeit(dgr, 20) if rc2 == 0 else float("nan")
        row.append("cfg%d fwd_xf %.1f us (%.0f TF)  dgrad_gx %.1f us" % (cfg, t1, 2.0 * rows * cin * cout / t1 * 1e-6, t2))
    print("\n   ".join(row))
raw.pn2_debug_set(8, 0)
