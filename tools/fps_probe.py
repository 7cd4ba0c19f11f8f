"""FPS timing probes (SA1 shape): variants of pn2_debug_set(0, v); v >= 10 are timing-only probes (wrong results)."""
import ctypes, sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.fps_ab import scene, timeit
L = ctypes.CDLL(os.path.abspath(sys.argv[1]))
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
shapes = [(16, 8192, 1024), (16, 4096, 512)]
for var in [int(a) for a in sys.argv[2:]]:
    L.pn2_debug_set(0, var)
    row = ["variant %3d" % var]
    for (b, n, m) in shapes:
        x = torch.from_numpy(scene(n, b, n)).cuda()
        out = torch.empty((b, m), dtype=torch.int32, device="cuda")
        f = lambda: L.pn2_farthest_point_sample(b, n, m, P(x), None, P(out), 1, st)
        assert f() == 0
        t = timeit(f)
        row.append("fps%s=%.1fus (%.0f ns/round)" % ((b, n, m), t, t / (m - 1) * 1e3))
    print("  ".join(row))
L.pn2_debug_set(0, 0)
