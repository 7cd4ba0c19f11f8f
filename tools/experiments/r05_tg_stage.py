import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, pn2_amd as pn2
dev = torch.device("cuda:0")
S = pn2.tf_ops.tf_sampling
def gtime(fn, iters=30):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
P = lambda t: ctypes.c_void_p(t.data_ptr())
for b, n, m in ((16, 8192, 1024), (16, 1024, 256)):
    x = torch.from_numpy(np.ascontiguousarray(bench.s_scene(7, b, n)[..., :3])).to(dev)
    _, known = S.farthest_point_sample_and_gather(m, x)
    d = torch.empty(b, n, 3, device=dev); i = torch.empty(b, n, 3, dtype=torch.int32, device=dev)
    dref, iref = pn2.three_nn(x, known, kernel=1)
    for v in sys.argv[1:]:
        L = ctypes.CDLL(os.path.join(ROOT, "tools/dbg/tg/libtg_%s.so" % v))
        f = lambda: L.tg_launch(b, n, m, P(x), 3, P(known), P(d), P(i), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        t = gtime(f)
        same = float((i == iref).float().mean())
        print("b=%d n=%d m=%d variant %-8s %.1f us   idx agreement with all-pairs %.4f" % (b, n, m, v, t, same), flush=True)
