"""Ball query / three_nn timing experiments (direct ctypes)."""
import ctypes, sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.fps_ab import scene, timeit  # noqa

def main():
    L = ctypes.CDLL(os.environ.get("PN2_HIP_LIBRARY") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "open3d-pointnet2-semantic3d_amd", "libpn2_hip.so"))
    dev = torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    b, n, m, ns = 16, 8192, 1024, 32
    x = torch.from_numpy(scene(0, b, n)).to(dev)
    f = torch.empty((b, m), dtype=torch.int32, device=dev)
    L.pn2_farthest_point_sample(b, n, m, P(x), None, P(f), 1, st)
    q = torch.empty((b, m, 3), device=dev)
    L.pn2_gather_point(b, n, m, P(x), P(f), P(q), st)
    idx = torch.empty((b, m, ns), dtype=torch.int32, device=dev)
    cnt = torch.empty((b, m), dtype=torch.int32, device=dev)
    for qpw in (8, 2):
        L.pn2_debug_set(1, qpw)
        for r in (0.5, 1e-3, 3.0):
            fn = lambda: L.pn2_query_ball_point(b, n, m, ctypes.c_float(r), ns, P(x), P(q), P(idx), P(cnt), 1, st)
            assert fn() == 0
            print("ball_query qpw_knob=%d r=%g: %.1f us  (mean cnt %.1f)" % (qpw, r, timeit(fn, 20), cnt.float().mean().item()))
    # fewer candidates / queries to see scaling
    L.pn2_debug_set(1, 8)
    for (nn, mm) in ((4096, 1024), (8192, 512), (8192, 2048)):
        xx = x[:, :nn].contiguous(); qq = q[:, :min(mm, m)].contiguous() if mm <= m else torch.cat([q, q], 1).contiguous()
        ii = torch.empty((b, qq.shape[1], ns), dtype=torch.int32, device=dev); cc = torch.empty((b, qq.shape[1]), dtype=torch.int32, device=dev)
        fn = lambda: L.pn2_query_ball_point(b, nn, qq.shape[1], ctypes.c_float(0.5), ns, P(xx), P(qq), P(ii), P(cc), 1, st)
        assert fn() == 0
        print("ball_query n=%d m=%d: %.1f us" % (nn, qq.shape[1], timeit(fn, 20)))
    for (nn, mm) in ((64, 1024), (256, 1024), (1024, 1024), (64, 64)):
        xx = x[:, :nn].contiguous(); qq = q[:, :mm].contiguous()
        ii = torch.empty((b, mm, ns), dtype=torch.int32, device=dev); cc = torch.empty((b, mm), dtype=torch.int32, device=dev)
        fn = lambda: L.pn2_query_ball_point(b, nn, mm, ctypes.c_float(0.5), ns, P(xx), P(qq), P(ii), P(cc), 1, st)
        assert fn() == 0
        print("ball_query n=%d m=%d: %.1f us" % (nn, mm, timeit(fn, 50)))
    g = torch.empty((b, 64, 3), device=dev)
    fn = lambda: L.pn2_gather_point(b, n, 64, P(x), P(f), P(g), st)
    print("gather_point tiny (host-overhead probe): %.1f us" % timeit(fn, 50))
    d = torch.empty((b, n, 3), device=dev); i3 = torch.empty((b, n, 3), dtype=torch.int32, device=dev)
    fn = lambda: L.pn2_three_nn(b, n, m, P(x), P(q), P(d), P(i3), st)
    assert fn() == 0
    print("three_nn n=8192 m=1024: %.1f us" % timeit(fn, 20))

if __name__ == "__main__":
    main()
