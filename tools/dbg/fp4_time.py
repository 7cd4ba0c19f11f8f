import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import pn2_amd as pn2
sys.path.insert(0, "/root/repo/tests")
from conftest import s_scene
dev = torch.device("cuda:0")
tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
B, N, M = 16, 8192, 1024
pc = s_scene(3, B, N)[:, :, :3]
xyz1 = torch.from_numpy(pc).to(dev)
xyz2 = pn2.gather_point(xyz1, pn2.farthest_point_sample(M, xyz1))
p1 = torch.rand(B, N, 3, device=dev); p2 = torch.randn(B, M, 128, device=dev)
dist, idx = pn2.three_nn(xyz1, xyz2)
tfu.set_default_store(tfu.VariableStore(device=dev, seed=1))
with tfu.variable_scope("fp"):
    f = lambda: pu.fp_features_inference(dist, idx, p1, p2, [128, 128, 128])
    out = f()
    for _ in range(30): f()
    pn2._lib.lib.trace = []
    for _ in range(30): f()
    torch.cuda.synchronize()
    tr, pn2._lib.lib.trace = pn2._lib.lib.trace, None
agg = {}
for nm, args, s_, e_ in tr: agg.setdefault(nm, []).append(s_.elapsed_time(e_) * 1e3)
print(os.environ.get("PN2_HIP_LIBRARY", "default")[-16:], " ".join("%s %.1f us" % (k, sum(v) / len(v)) for k, v in agg.items()), "checksum %.6f" % float(out.double().sum()))
