"""BASELINE configs[4] stage times: one scene, N=65536 -> npoint 4096, K=64, C=128 bf16 features."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import pn2_amd as pn2
from fps_ab import timeit
from conftest import s_scene
dev = torch.device("cuda:0")
N, M, K, C = 65536, 4096, 64, 128
xyz = torch.from_numpy(s_scene(0, 1, N)).to(dev)
pts = torch.randn(1, N, C, device=dev).to(torch.bfloat16)
t = timeit(lambda: pn2.tf_ops.tf_sampling.farthest_point_sample_and_gather(M, xyz), 3)
print("FPS %d -> %d: %.1f us (%.2f us per round)" % (N, M, t, t / (M - 1)))
_, new_xyz = pn2.tf_ops.tf_sampling.farthest_point_sample_and_gather(M, xyz)
t = timeit(lambda: pn2.query_ball_point(0.5, K, xyz, new_xyz), 10)
print("ball query r=0.5 K=%d: %.1f us" % (K, t))
idx, _ = pn2.query_ball_point(0.5, K, xyz, new_xyz)
tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
tfu.set_default_store(tfu.VariableStore(device=dev, seed=0))
with tfu.variable_scope("sa"):
    f = lambda: pu.sa_features_inference(xyz, new_xyz, pts, idx, [128, 128])
    f()
    t = timeit(f, 10)
print("fused bf16 MLP [128,128] + max: %.1f us" % t)
