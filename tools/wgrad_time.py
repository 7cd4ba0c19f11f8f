"""pn2_linear_wgrad timing at the training step's layer shapes (no tuning hooks needed)"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.fps_ab import timeit
lib = ctypes.CDLL(os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "open3d-pointnet2-semantic3d_amd", "libpn2_hip.so"))
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
tot = 0.0
for rows, cin, cout in [(524288, 6, 32), (524288, 32, 32), (524288, 32, 64), (131072, 67, 64), (131072, 64, 64), (131072, 64, 128),
                        (32768, 131, 128), (32768, 128, 128), (32768, 128, 256), (8192, 259, 256), (8192, 256, 256), (8192, 256, 512),
                        (1024, 768, 256), (1024, 256, 256), (4096, 384, 256), (4096, 256, 256), (16384, 320, 256), (16384, 256, 128),
                        (131072, 131, 128), (131072, 128, 128), (131072, 128, 128), (131072, 128, 128), (131072, 128, 9)]:
    x = torch.randn(rows, cin, device="cuda"); dy = torch.randn(rows, cout, device="cuda"); dw = torch.empty(cin, cout, device="cuda")
    f = lambda: lib.pn2_linear_wgrad(rows, cin, cout, P(x), P(dy), P(dw), st)
    assert f() == 0
    t = timeit(f, 10)
    err = float((dw - x.t() @ dy).abs().max() / (x.t() @ dy).abs().max())
    tot += t
    print("%7d x %3d -> %3d: %6.1f us  %5.1f TF  %5.2f TB/s  rel err %.1e" % (rows, cin, cout, t, 2.0 * rows * cin * cout / t * 1e-6, rows * (cin + cout) * 4 / t * 1e-6, err))
print("total %.1f us" % tot)
