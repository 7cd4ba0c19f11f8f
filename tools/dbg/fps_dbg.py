import ctypes, sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.fps_ab import scene
L = ctypes.CDLL(os.path.abspath(sys.argv[1]))
va, vb = int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
for (b, n, m) in [(1, 1024, 64), (2, 8192, 256), (16, 8192, 1024), (3, 5003, 700), (2, 2048, 2048), (2, 4096, 8)]:
    x = torch.from_numpy(scene(n, b, n)).to(dev)
    outs = []
    for var in (va, vb):
        L.pn2_debug_set(0, var)
        out = torch.full((b, m), -7, dtype=torch.int32, device=dev)
        assert L.pn2_farthest_point_sample(b, n, m, P(x), None, P(out), 1, st) == 0
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    a, c = outs
    bad = np.argwhere(a != c)
    print((b, n, m), "mismatches", len(bad), "first", bad[:1].tolist(), " new head", c[0, :6].tolist())
