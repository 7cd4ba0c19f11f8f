"""group_point tuning sweep at the north-star shape (B16 N8192 M1024 K32 C128)."""
import ctypes, sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.fps_ab import timeit
L = ctypes.CDLL(os.environ.get("PN2_HIP_LIBRARY") or os.path.join(ROOT, "open3d-pointnet2-semantic3d_amd", "libpn2_hip.so"))
dev = torch.device("cuda:0"); st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
b, n, m, k, c = 16, 8192, 1024, 32, 128
pts = torch.randn(b, n, c, device=dev)
idx = torch.randint(0, n, (b, m, k), dtype=torch.int32, device=dev)
out = torch.empty(b, m, k, c, device=dev)
byts = b * m * k * 4 + b * n * c * 4 + b * m * k * c * 4
ref = None
for var in (0, 2, 0, 2, 1, 3):
    L.pn2_debug_set(3, var)
    f = lambda: L.pn2_group_point(b, n, c, m, k, P(pts), P(idx), P(out), st)
    assert f() == 0
    t = timeit(f, 20)
    o = out.clone()
    if ref is None: ref = o
    assert torch.equal(ref, o)
    print("variant plain=%d noremap=%d blocks/CU=%d: %.1f us  %.0f GB/s  frac %.3f" % (var & 1, (var >> 1) & 1, var >> 4, t, byts / t / 1e3, byts / t / 1e3 / 8000))
# pure copy reference (same bytes written)
src = torch.empty_like(out)
t = timeit(lambda: out.copy_(src), 20)
print("torch copy 268MB->268MB: %.1f us  %.0f GB/s (read+write)" % (t, 2 * out.numel() * 4 / t / 1e3))
