"""Does the FP4 chain kernel (pn2_fp_mlp_fused_pre) speed up when the rows of a tile are spatial neighbours?  Same scene,
queries in input order vs sorted along a Morton curve (so that 32 consecutive rows share their three_nn neighbours and the
gathered rows of z hit the vector L1).  usage: python tools/fp4_order_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pn2_amd as pn2
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import s_scene
dev = torch.device("cuda:0")
tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
B, N, M = 16, 8192, 1024


def morton(x):
    q = ((x - x.min(1, keepdims=True)) / (np.ptp(x, 1, keepdims=True) + 1e-9) * 1023).astype(np.uint64)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        return (v | (v << 2)) & 0x09249249
    return spread(q[..., 0]) | (spread(q[..., 1]) << 1) | (spread(q[..., 2]) << 2)


pc = s_scene(3, B, N)[:, :, :3]
for name in ("input order", "morton order"):
    x = pc.copy()
    if name == "morton order":
        o = np.argsort(morton(x), axis=1)
        x = np.take_along_axis(x, o[:, :, None], 1)
    xyz1 = torch.from_numpy(x).to(dev)
    fps = pn2.farthest_point_sample(M, xyz1)
    xyz2 = pn2.gather_point(xyz1, fps)
    p1 = torch.rand(B, N, 3, device=dev)
    p2 = torch.randn(B, M, 128, device=dev)
    dist, idx = pn2.three_nn(xyz1, xyz2)
    tfu.set_default_store(tfu.VariableStore(device=dev, seed=1))
    with tfu.variable_scope("fp"):
        f = lambda: pu.fp_features_inference(dist, idx, p1, p2, [128, 128, 128])
        for _ in range(3):
            f()
        pn2._lib.lib.trace = []
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        tr, pn2._lib.lib.trace = pn2._lib.lib.trace, None
    agg = {}
    for nm, args, s_, e_ in tr:
        agg.setdefault(nm, []).append(s_.elapsed_time(e_) * 1e3)
    print(name, " ".join("%s %.1f us" % (k, sum(v) / len(v)) for k, v in agg.items()))
