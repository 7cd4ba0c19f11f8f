"""InterpolateLabelWithColor timing: device grid kNN vs a CPU KD-tree (scipy cKDTree, all cores) standing in for the
reference's Open3D/FLANN + OpenMP path (tf_interpolate.cpp:71-115).  usage: python tools/label_ab.py [ns nd]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pn2_amd as pn2
ns, nd = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000000, 10000000)
rs = np.random.RandomState(0)
def cloud(n):  # 100 m x 100 m scene, points near a ground surface + some structures
    return np.concatenate([rs.uniform(-50, 50, (n, 2)), np.abs(rs.normal(0, 2.0, (n, 1)))], 1).astype(np.float32)
sp, dp = cloud(ns), cloud(nd)
sl = rs.randint(0, 9, ns).astype(np.int32)
dev = torch.device("cuda:0")
tsp, tsl, tdp = torch.from_numpy(sp).to(dev), torch.from_numpy(sl).to(dev), torch.from_numpy(dp).to(dev)
for _ in range(2):
    lab, col = pn2.interpolate_label_with_color(tsp, tsl, tdp, 3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    lab, col = pn2.interpolate_label_with_color(tsp, tsl, tdp, 3)
torch.cuda.synchronize()
gpu = (time.perf_counter() - t0) / 3
print("GPU  ns=%d nd=%d knn=3: %.1f ms  (%.1f M dense points/s)" % (ns, nd, gpu * 1e3, nd / gpu * 1e-6))
try:
    from scipy.spatial import cKDTree
    sub = min(nd, 1000000)
    t0 = time.perf_counter()
    tree = cKDTree(sp.astype(np.float64))
    _, ii = tree.query(dp[:sub].astype(np.float64), k=3, workers=-1)
    cpu = time.perf_counter() - t0
    print("CPU  cKDTree build + %d queries on %d cores: %.2f s  (%.2f M dense points/s)" % (sub, os.cpu_count(), cpu, sub / cpu * 1e-6))
    # spot-check the labels of the first 20000 points with the vote restated in numpy
    l3 = sl[ii[:20000]]
    ref = np.where((l3[:, 1] == l3[:, 2]) & (l3[:, 0] != l3[:, 1]), l3[:, 1], l3[:, 0])
    print("label agreement on 20000 points: %.5f" % (lab[:20000].cpu().numpy() == ref).mean())
except Exception as e:  # scipy missing on the box
    print("CPU leg skipped:", e)
