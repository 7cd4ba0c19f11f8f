"""pn2_coarse_geometry (levels 2-4 of configs[1] + the 3-NN tables of FP1-FP3, one launch) against the nine separate launches
it replaces, graph-timed; with a TUNING build (PN2_HIP_LIBRARY=tools/ab/libpn2_tune.so) sweeps the cost units per wave (hook 15).
usage: python tools/coarse_geometry_ab.py [units ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import pn2_amd as pn2
from conftest import s_grid, s_scene, s_randn
dev = torch.device("cuda:0")
pu, S = pn2.util.pointnet_util, pn2.tf_ops.tf_sampling
lib = pn2._lib.lib


def graph_time(fn, iters=50):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return float(np.median(ts))


for name, gen in (("S-scene", s_scene), ("S-randn", s_randn), ("16-lattice (every level through the sampler)", lambda s, b, n: s_grid(s, b, n, 16))):
    x = torch.from_numpy(gen(0, 16, 8192)).to(dev)
    _, l1 = S.farthest_point_sample_and_gather(1024, x)
    radii = [1.0, 2.0, 4.0] if name.startswith("S-") else [0.12, 0.24, 0.48]

    def separate():
        cur = l1
        for m, r in zip((256, 64, 16), radii):
            _, nx = S.farthest_point_sample_and_gather(m, cur)
            pn2.query_ball_point(r, 32, cur, nx)
            pn2.three_nn(cur, nx)
            cur = nx

    def merged():
        pu.coarse_geometry(l1, [256, 64, 16], radii, [32, 32, 32])

    print("%s: nine separate launches %.1f us" % (name, graph_time(separate)))
    units = [int(a) for a in sys.argv[1:]] or [0]
    for u in units:
        if u > 0:
            assert lib._raw.pn2_debug_set(15, u) == 0, "needs a tuning build"
        print("%s: pn2_coarse_geometry%s %.1f us" % (name, " units/wave %d" % u if u > 0 else "", graph_time(merged)))
