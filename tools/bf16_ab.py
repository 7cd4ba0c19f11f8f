"""fp32 vs bf16 fused grouped MLP at the configs[4] shape (B=1, N=65536, M=4096, K=64, C=128 -> 128 -> 128 + max) and at
the north-star shape with K=32.  usage: python tools/bf16_ab.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import pn2_amd as pn2
from fps_ab import timeit
raw = pn2._lib._raw
dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (b, n, m, K, c, mlp) in [(1, 65536, 4096, 64, 128, [128, 128]), (16, 8192, 1024, 32, 128, [128]), (16, 8192, 1024, 32, 128, [128, 128, 128])]:
    g = torch.Generator().manual_seed(0)
    xyz = torch.rand(b, n, 3, generator=g).to(dev)
    new_xyz = xyz[:, :m].contiguous()
    idx = torch.randint(0, n, (b, m, K), generator=g, dtype=torch.int32).to(dev)
    pts = torch.randn(b, n, c, generator=g).to(dev)
    ptsb = pts.to(torch.bfloat16)
    ws, bs, cin = [], [], 3 + c
    for w_ in mlp:
        ws.append((torch.randn(cin, w_, generator=g) / cin ** 0.5).to(dev)); bs.append(torch.zeros(w_, device=dev)); cin = w_
    L = len(mlp)
    widths = (ctypes.c_int * L)(*mlp)
    wp = (ctypes.c_void_p * L)(*[t.data_ptr() for t in ws]); bp = (ctypes.c_void_p * L)(*[t.data_ptr() for t in bs])
    out = torch.empty(b, m, mlp[-1], device=dev)
    flops = 2.0 * b * m * K * sum(a * o for a, o in zip([3 + c] + mlp[:-1], mlp))
    row = ["b%d n%d m%d K%d c%d %s" % (b, n, m, K, c, mlp)]
    f16 = lambda: raw.pn2_sa_mlp_max_fused_bf16(b, n, m, K, c, P(xyz), P(new_xyz), P(ptsb), P(idx), L, widths, wp, bp, P(out), st)
    assert f16() == 0
    t = timeit(f16, 20); row.append("bf16 %.1f us (%.0f TF)" % (t, flops / t * 1e-6))
    f32 = lambda: raw.pn2_sa_mlp_max_fused(b, n, m, K, c, P(xyz), P(new_xyz), P(pts), P(idx), L, widths, wp, bp, P(out), st)
    if f32() == 0:
        t = timeit(f32, 20); row.append("fp32 %.1f us (%.0f TF)" % (t, flops / t * 1e-6))
    else:
        row.append("fp32 fused: unsupported (K=%d)" % K)
    print("  ".join(row))
