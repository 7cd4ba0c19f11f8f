"""LDS-grid ball query stage breakdown: python tools/bq_stage_ab.py --build (CPU box), then run on the GPU."""
import ctypes, sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
PKG = os.path.join(ROOT, "open3d-pointnet2-semantic3d_amd")
STAGES = [1, 2, 3, 4, 9]
if "--build" in sys.argv:
    import importlib.util
    spec = importlib.util.spec_from_file_location("b", os.path.join(PKG, "build.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    for k in STAGES:
        b.build(force=True, extra_flags=["-DPN2_BQG_STAGE=%d" % k], out=os.path.join(PKG, "libpn2_bqg%d.so" % k))
    sys.exit(0)
import torch
from fps_ab import scene, timeit
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
b, n, m = 16, 8192, 1024
x = torch.from_numpy(scene(1, b, n)).to(dev); q = x[:, :m].contiguous()
idx = torch.empty((b, m, 32), dtype=torch.int32, device=dev); cnt = torch.empty((b, m), dtype=torch.int32, device=dev)
for k in STAGES:
    L = ctypes.CDLL(os.path.join(PKG, "libpn2_bqg%d.so" % k))
    L.pn2_query_ball_point_kernel.argtypes = [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    f = lambda: L.pn2_query_ball_point_kernel(b, n, m, 0.5, 32, P(x), P(q), P(idx), P(cnt), 1, 3, st)  # kernel 3 = LDS grid
    assert f() == 0
    print("stage<=%d: %.1f us" % (k, timeit(f, 30)))
