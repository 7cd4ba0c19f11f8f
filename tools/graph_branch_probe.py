"""Do independent branches of ONE captured hipGraph run concurrently on this stack?  Two FPS launches (16 workgroups, ~630 us
each, 16 of 256 CUs) captured on two streams inside one graph: replay ~630 us = concurrent, ~1260 us = serialised."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pn2_amd as pn2
from bench import s_scene
dev = torch.device("cuda:0")
x1 = torch.from_numpy(s_scene(0, 16, 8192)[:, :, :3].copy()).to(dev)
x2 = torch.from_numpy(s_scene(1, 16, 8192)[:, :, :3].copy()).to(dev)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
main = torch.cuda.Stream(); side = torch.cuda.Stream()
print("one FPS eager: %.0f us" % t(lambda: pn2.farthest_point_sample(1024, x1)))
def two_eager():
    with torch.cuda.stream(main): pn2.farthest_point_sample(1024, x1)
    with torch.cuda.stream(side): pn2.farthest_point_sample(1024, x2)
print("two FPS eager on two streams: %.0f us" % t(two_eager))
with torch.cuda.stream(main):
    pn2.farthest_point_sample(1024, x1); pn2.farthest_point_sample(1024, x2)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main):
        side.wait_stream(main)
        a = pn2.farthest_point_sample(1024, x1)
        with torch.cuda.stream(side):
            b = pn2.farthest_point_sample(1024, x2)
        main.wait_stream(side)
    print("graph with two branches, replay: %.0f us" % t(g.replay))
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=main):
        a = pn2.farthest_point_sample(1024, x1)
        b = pn2.farthest_point_sample(1024, x2)
    print("graph with the two in sequence, replay: %.0f us" % t(g2.replay))
