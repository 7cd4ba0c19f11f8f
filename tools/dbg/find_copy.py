"""which Python line of an inference forward launches an aten copy?  (torch profiler with stacks)"""
import os, sys
import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pn2_amd as pn2
dev = torch.device("cuda:0")
tfu = pn2.util.tf_util
hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
B, N = 4, 8192
hp.update(batch_size=B, num_point=N)
rs = np.random.RandomState(0)
pc = torch.from_numpy(rs.random_sample((B, N, 6)).astype(np.float32) * 5).to(dev)
tfu.set_default_store(tfu.VariableStore(device=dev, seed=2))
with torch.no_grad():
    pn2.model.get_sa_fp_features(pc, False, hp)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        pn2.model.get_sa_fp_features(pc, False, hp)
        torch.cuda.synchronize()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::cat", "aten::_to_copy") and e.device_type == torch.autograd.DeviceType.CPU:
        print(e.name, [s for s in (e.stack or []) if "repo" in s][:4])
