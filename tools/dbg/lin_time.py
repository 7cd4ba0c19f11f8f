"""time pn2_linear at a few shapes (graph of 20 launches, HIP events): python tools/dbg/lin_time.py"""
import sys, os
sys.path.insert(0, "/root/repo")
import torch
import pn2_amd as pn2
dev = torch.device("cuda:0")
tfu = pn2.util.tf_util
for rows, cin, cout in [(16384, 64, 64), (4096, 128, 128), (16384, 128, 128), (4096, 256, 256), (1024, 256, 256), (131072, 64, 64)]:
    x = torch.randn(rows, cin, device=dev); w = torch.randn(cin, cout, device=dev)
    f = lambda: tfu.hip_linear(x, w, None, relu=False)
    ref = (x.double() @ w.double())
    err = float((f().double() - ref).abs().max() / ref.abs().max())
    for _ in range(5): f()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    print("%s (%d,%d,%d) %.2f us  rel err %.1e" % (os.environ.get("PN2_HIP_LIBRARY", "default")[-14:], rows, cin, cout, a.elapsed_time(b) * 50, err))
