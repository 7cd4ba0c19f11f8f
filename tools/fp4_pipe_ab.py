"""A/B of the FP4 chain (pn2_fp_mlp_fused_pre at 16 x 8192 rows, 131 -> 128 -> 128 -> 128): lockstep kernel (schedule 0)
against the software-pipelined kernel (schedule 1), HIP events, same inputs; prints us per launch and the MFMA fraction
(executed flops 2 * rows * (3*128 + 128*128 + 128*128) / 157.3 TFLOP/s)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pn2_amd as pn2  # noqa: E402

dev = torch.device("cuda:0")
tfu = pn2.util.tf_util
rs = np.random.RandomState(0)
b, n, m, c1, c2 = 16, 8192, 1024, 3, 256
xy = rs.uniform(-5, 5, (b, n, 2)); z = np.clip(np.abs(rs.normal(0, 1.5, (b, n, 1))), 0, 8)
xyz1 = torch.from_numpy(np.concatenate([xy, z], 2).astype(np.float32)).to(dev)
_, xyz2 = pn2.tf_ops.tf_sampling.farthest_point_sample_and_gather(m, xyz1)
dist, idx = pn2.three_nn(xyz1, xyz2)
p1 = torch.from_numpy(rs.rand(b, n, c1).astype(np.float32)).to(dev)
p2 = torch.from_numpy(rs.randn(b, m, c2).astype(np.float32)).to(dev)
ws, bs, c = [], [], c1 + c2
for w_ in (128, 128, 128):
    ws.append(torch.from_numpy((rs.randn(c, w_) / np.sqrt(c)).astype(np.float32)).to(dev))
    bs.append(torch.from_numpy((rs.randn(w_) * 0.1).astype(np.float32)).to(dev))
    c = w_
flops = 2 * b * n * (c1 * 128 + 128 * 128 + 128 * 128)
pn2._lib.lib.trace = []
for sched in (0, 1, 0, 1):
    for _ in range(12):
        tfu.hip_fp_mlp_fused_pre(dist, idx, p1, p2, ws, bs, schedule=sched)
torch.cuda.synchronize()
tr, pn2._lib.lib.trace = pn2._lib.lib.trace, None
acc = {}
for name, args, s, e in tr:
    if name == "pn2_fp_mlp_fused_pre_schedule":
        acc.setdefault(args[5], []).append(s.elapsed_time(e) * 1e3)  # ints: b, n, m, c1, nlayers, schedule, widths...
for k, v in sorted(acc.items()):
    v = sorted(v)[2:-2]
    us = sum(v) / len(v)
    print("schedule/key", k, "n", len(v), "avg_us %.2f min %.2f" % (us, v[0]), "frac %.3f" % (flops / (us * 1e-6) / 157.3e12))
