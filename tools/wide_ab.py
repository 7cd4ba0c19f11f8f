"""pn2_mlp_wide (one launch per coarse level) against one pn2_linear per layer at the SSG model's coarse-level shapes."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
def timeit(fn, iters=20):
    """kernel time, not launch time: `iters` calls captured into one hipGraph (the Python wrappers cost 10-40 us per call,
    more than these kernels run)"""
    for _ in range(2): fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(iters): fn()
        g.replay(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(st); g.replay(); g.replay(); g.replay(); e.record(st); torch.cuda.synchronize()
    return s.elapsed_time(e) / (3 * iters) * 1e3
import pn2_amd as pn2
tfu = pn2.util.tf_util
dev = torch.device("cuda:0")
shapes = [("SA4 (materialised input)", 8192, 259, (256, 256, 512), 32), ("FP3", 16384, 320, (256, 128), 0), ("FP2", 4096, 384, (256, 256), 0),
          ("SA3 tail", 32768, 128, (256,), 32), ("FP1", 1024, 768, (256, 256), 0)]
for name, rows, cin, widths, pool in shapes:
    x = torch.randn(rows, cin, device=dev)
    ws, bs, c = [], [], cin
    for w_ in widths:
        ws.append(torch.randn(c, w_, device=dev) / c ** 0.5); bs.append(torch.randn(w_, device=dev) * 0.1); c = w_
    def per_layer():
        h = x
        for i, (w_, b_) in enumerate(zip(ws, bs)):
            h = tfu.hip_linear(h, w_, b_, relu=True, pool=pool if i == len(ws) - 1 else 0)
        return h
    wide = lambda: tfu.hip_mlp_wide(x, ws, bs, pool=pool)
    a, b_ = per_layer(), wide()
    err = float((a - b_).abs().max())
    flops = 2.0 * rows * sum(ci * co for ci, co in zip((cin,) + widths[:-1], widths))
    t1, t2 = timeit(per_layer, 30), timeit(wide, 30)
    print("%-26s rows %6d %4d->%s  per-layer %6.1f us (%3.0f TF)  wide %6.1f us (%3.0f TF)  max|diff| %.1e"
          % (name, rows, cin, "->".join(map(str, widths)), t1, flops / t1 * 1e-6, t2, flops / t2 * 1e-6, err))
# SA4 with the gather front end
b, n, m, c = 16, 64, 16, 256
xyz = torch.rand(b, n, 3, device=dev); new_xyz = xyz[:, :m].contiguous(); pts = torch.randn(b, n, c, device=dev)
idx = torch.randint(0, n, (b, m, 32), device=dev, dtype=torch.int32)
ws, bs, cc = [], [], 3 + c
for w_ in (256, 256, 512):
    ws.append(torch.randn(cc, w_, device=dev) / cc ** 0.5); bs.append(torch.randn(w_, device=dev) * 0.1); cc = w_
pu = pn2.util.pointnet_util
def unfused():
    h = pu._sa_group_concat(xyz, new_xyz, pts, idx).reshape(-1, 3 + c)
    for i, (w_, b_) in enumerate(zip(ws, bs)):
        h = tfu.hip_linear(h, w_, b_, relu=True, pool=32 if i == 2 else 0)
    return h
kws = [tfu.sa_wide_first_layer(ws[0])] + ws[1:]
fused = lambda: tfu.hip_sa_mlp_wide(xyz, new_xyz, pts, idx, kws, bs)
print("SA4 gather+MLP+max: group_concat + 3 x pn2_linear %.1f us, pn2_sa_mlp_wide %.1f us, max|diff| %.1e"
      % (timeit(unfused, 30), timeit(fused, 30), float((unfused().reshape(b, m, -1) - fused()).abs().max())))
