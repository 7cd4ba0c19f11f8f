"""BASELINE configs[2] (MSG SA, 3 scales, B=16, N=8192, M=1024): multi-radius ball query vs three scans, and the whole
module (captured in a hipGraph) fused vs unfused.  usage: python tools/msg_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import pn2_amd as pn2
from fps_ab import timeit
from conftest import s_scene
dev = torch.device("cuda:0")
B, N, M = 16, 8192, 1024
radii, ks, mlps = [0.25, 0.5, 1.0], [16, 32, 64], [[32, 32, 64], [64, 64, 128], [64, 96, 128]]
xyz = torch.from_numpy(s_scene(0, B, N)).to(dev)
pts = torch.rand(B, N, 3, device=dev)
_, new_xyz = pn2.tf_ops.tf_sampling.farthest_point_sample_and_gather(M, xyz)
g = pn2.tf_ops.tf_grouping
t_multi = timeit(lambda: g.query_ball_point_multi(radii, ks, xyz, new_xyz), 20)
t_sep = timeit(lambda: [g.query_ball_point(r, k, xyz, new_xyz) for r, k in zip(radii, ks)], 20)
print("ball query x3 radii: one scan %.1f us, three scans %.1f us" % (t_multi, t_sep))
tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
tfu.set_default_store(tfu.VariableStore(device=dev, seed=1))
def fwd(x):
    with torch.no_grad():
        return pu.pointnet_sa_module_msg(x, pts, M, radii, ks, mlps, False, None, scope="msg")[1]
for fused in (True, False):
    pu.USE_FUSED_SA = fused
    cap = pn2.runtime.CapturedForward(fwd, xyz)
    t = timeit(lambda: cap.replay(), 20)
    print("MSG module (graph replay) fused=%s: %.1f us (%.1f M points/s)" % (fused, t, B * N / t))
pu.USE_FUSED_SA = True
