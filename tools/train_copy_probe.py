import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import pn2_amd as pn2
from bench import s_scene
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
B, N = 16, 8192
hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
rs = np.random.RandomState(100)
pc = torch.from_numpy(np.concatenate([s_scene(3000, B, N), rs.random_sample((B, N, 3)).astype(np.float32)], 2)).to(dev)
labels = torch.from_numpy(rs.randint(0, 9, (B, N)).astype(np.int64)).to(dev)
smpw = torch.from_numpy((rs.random_sample((B, N)) + 0.5).astype(np.float32)).to(dev)
tr = pn2.train.Trainer(hp, 9, store=pn2.util.tf_util.VariableStore(device=dev, seed=0), device=dev, capture=False)
for i in range(3):
    tr.train_step(pc, labels, smpw, next_pc=pc)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.train_step(pc, labels, smpw, next_pc=pc)
    torch.cuda.synchronize()
from collections import Counter
c = Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::contiguous", "aten::clone", "aten::cat", "aten::add", "aten::add_", "aten::mul", "aten::sum", "aten::select_backward", "aten::slice_backward"):
        st = [s for s in (ev.stack or []) if "pn2" in s or "open3d" in s or "train.py" in s or "model.py" in s or "tf_util" in s or "pointnet_util" in s][:2]
        c[(ev.name, str(ev.input_shapes)[:70], " <- ".join(x.split("/")[-1][:60] for x in st))] += 1
for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:45]:
    print(v, k)
