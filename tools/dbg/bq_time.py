import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pn2_amd as pn2
from bench import s_scene, time_call
dev = torch.device("cuda:0")
g = pn2.tf_ops.tf_grouping
for (b, n, m, r, K) in [(16, 8192, 1024, 0.5, 32), (16, 4096, 512, 0.7, 32), (16, 8192, 1024, 1.0, 64)]:
    x = torch.from_numpy(s_scene(0, b, n)[:, :, :3].copy()).to(dev)
    nx = pn2.gather_point(x, pn2.farthest_point_sample(m, x))
    t0 = time_call(lambda: pn2.query_ball_point(r, K, x, nx), 20)
    ws = g.ball_query_bin_alloc(x)
    tb = time_call(lambda: g.ball_query_bin(r, x, out=ws), 20)
    t1 = time_call(lambda: g.query_ball_point_binned(r, K, x, nx, ws), 20)
    print("b=%d n=%d m=%d r=%.1f K=%d: unbinned %.1f us | bin %.1f us + binned query %.1f us" % (b, n, m, r, K, t0 * 1e3, tb * 1e3, t1 * 1e3))
