"""three_nn stage breakdown: build truncated variants (-DPN2_NN_STAGES=k) on the CPU box first
(python tools/nn_stage_ab.py --build), then time them on the GPU (python tools/nn_stage_ab.py)."""
import ctypes, sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "open3d-pointnet2-semantic3d_amd")
STAGES = [1, 2, 3, 4, 9]
if "--build" in sys.argv:
    import importlib.util
    spec = importlib.util.spec_from_file_location("b", os.path.join(PKG, "build.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    for k in STAGES:
        b.build(force=True, extra_flags=["-DPN2_NN_STAGES=%d" % k], out=os.path.join(PKG, "libpn2_nn%d.so" % k))
    sys.exit(0)
import torch
from fps_ab import scene, timeit
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
for k in STAGES:
    L = ctypes.CDLL(os.path.join(PKG, "libpn2_nn%d.so" % k))
    row = ["stages<=%d" % k]
    for (b, n, m) in [(16, 8192, 1024), (16, 1024, 256), (1, 65536, 4096)]:
        a = torch.from_numpy(scene(1, b, n)).to(dev); r = a[:, :m].contiguous()
        d = torch.empty((b, n, 3), device=dev); i = torch.empty((b, n, 3), dtype=torch.int32, device=dev)
        f = lambda: L.pn2_three_nn(b, n, m, P(a), P(r), P(d), P(i), st)
        assert f() == 0
        row.append("(%d,%d,%d)=%.1fus" % (b, n, m, timeit(f, 20)))
    print("  ".join(row))
