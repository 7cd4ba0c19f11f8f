"""ball query kernels at the SA1 shape: lane (scan) kernel vs LDS-grid kernel; and the grid kernel's fixed build cost
(m = 64 queries per batch element: one workgroup per element, so time ~= build + one query pass)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import pn2_amd as pn2
from fps_ab import timeit
from conftest import s_scene
raw = pn2._lib._raw
dev = torch.device("cuda:0")
for (b, n, m, r, K) in [(16, 8192, 1024, 0.5, 32), (16, 8192, 64, 0.5, 32), (16, 8192, 1024, 1.0, 64), (16, 8192, 1024, 0.25, 16), (16, 4096, 512, 0.5, 32)]:
    xyz = torch.from_numpy(s_scene(0, b, n)).to(dev)
    _, q = pn2.tf_ops.tf_sampling.farthest_point_sample_and_gather(m, xyz)
    row = ["b%d n%d m%d r%g K%d" % (b, n, m, r, K)]
    for name, var in (("lane", 2), ("grid", 3)):
        raw.pn2_debug_set(2, var)
        t = timeit(lambda: pn2.query_ball_point(r, K, xyz, q), 30)
        row.append("%s %.1f us" % (name, t))
    raw.pn2_debug_set(2, 0)
    print("  ".join(row))
