"""dependent-launch cost inside a replayed hipGraph: N tiny kernels in a chain"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pn2_amd as pn2
from pn2_amd._lib import lib, ptr, stream_ptr
dev = torch.device("cuda:0")
z = torch.randn(256, device=dev); dz = torch.randn(256, device=dev); dx = torch.empty(256, device=dev)
for N in (1, 21, 100):
    for nel in (256, 1 << 20):
        z = torch.randn(nel, device=dev); dz = torch.randn(nel, device=dev); dx = torch.empty(nel, device=dev)
        f = lambda: lib.pn2_relu_grad(nel, ptr(z), ptr(dz), ptr(dx), stream_ptr())
        f(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(N): f()
        g.replay(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            s.record(); g.replay(); e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) * 1e3)
        print("chain of %3d x relu_grad(%7d elements): %.1f us per replay = %.2f us per launch" % (N, nel, best, best / N))
