"""Training-step time at BASELINE configs[3]'s per-GPU shape (B=16, N=8192, semantic.json, fp32): forward with batch-stat BN,
weighted CE, backward through the HIP gradient kernels, flat all-reduce (world 1), Adam.  usage: python tools/train_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import pn2_amd as pn2
from conftest import s_scene
dev = torch.device("cuda:0")
B, N = 16, 8192
rs = np.random.RandomState(0)
pc = torch.from_numpy(np.concatenate([s_scene(0, B, N), rs.random_sample((B, N, 3)).astype(np.float32)], 2)).to(dev)
labels = torch.from_numpy(rs.randint(0, 9, (B, N)).astype(np.int64)).to(dev)
smpw = torch.from_numpy((rs.random_sample((B, N)) + 0.5).astype(np.float32)).to(dev)
tr = pn2.train.Trainer(pn2.model.SEMANTIC_HYPERPARAMS, 9, store=pn2.util.tf_util.VariableStore(device=dev, seed=3))
for _ in range(3):
    tr.train_step(pc, labels, smpw)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for _ in range(K):
    loss = tr.train_step(pc, labels, smpw)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("train step B=%d N=%d: %.2f ms  (%.2f M points/s), loss %.4f" % (B, N, dt * 1e3, B * N / dt * 1e-6, loss))
# where the time goes
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    tr.train_step(pc, labels, smpw)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
