"""Which launches of one (eager) training step are NOT the library's kernels, and which Python line issues each.

    gpurun -- 'python tools/train_glue_probe.py > gpurun_out/glue.txt'

torch profiler over one eager step after warm-up (the captured step records the same launches).  For every device kernel whose
name is not one of libpn2_hip.so's (anonymous-namespace kernels), print the aten op that launched it, its input shapes and
the innermost frames of this package on the Python stack; then the per-kernel-name totals of the step."""
import os
import sys
from collections import Counter, defaultdict

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import pn2_amd as pn2  # noqa: E402
from bench import s_scene  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

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

MINE = ("pn2", "open3d", "train.py", "model.py", "tf_util", "pointnet_util", "dist.py", "tf_sampling", "tf_grouping", "tf_interpolate")


def frames(ev):
    st = [s for s in (ev.stack or []) if any(m in s for m in MINE)][:3]
    return " <- ".join(x.split("/")[-1][:70] for x in st)


ops = Counter()
dur = defaultdict(float)
names = Counter()
kdur = defaultdict(float)
for ev in prof.events():
    ks = getattr(ev, "kernels", None) or []
    if ev.device_type == torch.autograd.DeviceType.CPU and ks:
        for k in ks:
            lib_kernel = "anonymous namespace" in k.name and "at::" not in k.name
            names[k.name[:90]] += 1
            kdur[k.name[:90]] += k.duration
            if not lib_kernel:
                key = (ev.name, str(ev.input_shapes)[:80], frames(ev), k.name[:60])
                ops[key] += 1
                dur[key] += k.duration
print("== launches that are not the library's kernels (one eager step) ==")
for k, v in sorted(ops.items(), key=lambda kv: -dur[kv[0]]):
    print("%3d x %7.1f us  %s" % (v, dur[k], k))
print("total: %d launches, %.1f us" % (sum(ops.values()), sum(dur.values())))
print("== all kernels of the step ==")
for k, v in sorted(names.items(), key=lambda kv: -kdur[kv[0]]):
    print("%3d x %8.1f us  %s" % (v, kdur[k], k))
print("total kernels %d, %.1f us" % (sum(names.values()), sum(kdur.values())))
