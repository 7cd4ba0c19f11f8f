"""pn2_linear timing at the model's layer shapes for register prefetch depths 2/3/4 (pn2_debug_set(5, st))."""
import ctypes, sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from fps_ab import timeit
import pn2_amd as pn2
raw = pn2._lib._raw
dev = torch.device("cuda:0")
shapes = [(16384, 320, 256, 0), (8192, 256, 512, 32), (32768, 128, 256, 32), (8192, 259, 256, 0), (8192, 256, 256, 0),
          (1024, 768, 256, 0), (16384, 256, 128, 0), (4096, 384, 256, 0), (1024, 256, 256, 0), (4096, 256, 256, 0),
          (524288, 128, 128, 32), (131072, 128, 128, 0)]
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (rows, cin, cout, pool) in shapes:
    x = torch.randn(rows, cin, device=dev); w = torch.randn(cin, cout, device=dev) / cin ** 0.5; b = torch.randn(cout, device=dev)
    y = torch.empty(rows // pool if pool else rows, cout, device=dev)
    row = ["(%d,%d,%d,p%d)" % (rows, cin, cout, pool)]
    ref = None
    for depth in (0, 1, 2, 3, 4, 5, 6, 7, 8):
        raw.pn2_debug_set(8, depth)
        f = lambda: raw.pn2_linear(rows, cin, cout, P(x), P(w), P(b), 1, pool, P(y), st)
        assert f() == 0
        t = timeit(f, 30)
        o = y.clone()
        if ref is None: ref = o
        assert torch.allclose(ref, o, rtol=1e-5, atol=1e-5)
        row.append("cfg%d=%.1fus(%.0fTF)" % (depth, t, 2.0 * rows * cin * cout / t * 1e-6))
    print("  ".join(row))
raw.pn2_debug_set(8, 0)
