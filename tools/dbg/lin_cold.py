"""SA3's pooled linear [32768,128,256,pool 32]: 21 us when timed back to back on a resident input, 36-47 us in bench.py's eager
bracket.  Is it the input just written by the producing kernel (cold in this XCD's L2)?"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pn2_amd as pn2
from pn2_amd._lib import lib, ptr, stream_ptr
dev = torch.device("cuda:0")
rows, cin, cout, pool = 32768, 128, 256, 32
x = torch.randn(rows, cin, device=dev); w = torch.randn(cin, cout, device=dev) / 11; b = torch.randn(cout, device=dev)
y = torch.empty(rows // pool, cout, device=dev)
src = torch.randn(rows, cin, device=dev)
big = torch.empty(64 << 20, device=dev)  # 256 MB: flushes L2 + most of the Infinity Cache


def gtime(fns, iters=30):
    for f in fns: f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            for f in fns: f()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


lin = lambda: lib.pn2_linear(rows, cin, cout, ptr(x), ptr(w), ptr(b), 1, pool, ptr(y), stream_ptr())
wr = lambda: lib.pn2_relu_grad(x.numel(), ptr(src), ptr(src), ptr(x), stream_ptr())   # rewrites x (16.8 MB) from another buffer
fl = lambda: big.fill_(1.0)
t_lin, t_wr, t_pair = gtime([lin]), gtime([wr]), gtime([wr, lin])
t_fl, t_flpair = gtime([fl], 5), gtime([fl, lin], 5)
print("linear alone %.1f us | writer alone %.1f | writer + linear %.1f -> linear after a fresh write %.1f us" % (t_lin, t_wr, t_pair, t_pair - t_wr))
print("256 MB fill alone %.1f | fill + linear %.1f -> linear on an input evicted to HBM %.1f us" % (t_fl, t_flpair, t_flpair - t_fl))
