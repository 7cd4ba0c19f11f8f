"""single GPU: cost of the multi-rank capture forms (forced): one graph | two graphs | three graphs.  python tools/dbg/train_split_time.py"""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import pn2_amd as pn2
from conftest import s_scene
dev = torch.device("cuda:0")
hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
B, N = 16, 8192
rs = np.random.RandomState(0)
def batch(seed):
    pc = torch.from_numpy(np.concatenate([s_scene(seed, B, N)[:, :, :3], rs.rand(B, N, 3).astype(np.float32)], 2)).to(dev)
    return pc, torch.from_numpy(rs.randint(0, 9, (B, N))).to(dev), torch.from_numpy((rs.rand(B, N) + 0.5).astype(np.float32)).to(dev)
bs = [batch(s) for s in range(3)]
for name, kw in (("one graph", dict(split_capture=False)), ("two graphs", dict(split_capture=True, overlap_collective=False)),
                 ("three graphs", dict(split_capture=True, overlap_collective=True))):
    tr = pn2.train.Trainer(hp, 9, store=pn2.util.tf_util.VariableStore(device=dev, seed=1), warmup_eager=2, **kw)
    for i in range(8):
        tr.train_step(*bs[i % 3], sync=False, next_pc=bs[(i + 1) % 3][0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(30):
        tr.train_step(*bs[i % 3], sync=False, next_pc=bs[(i + 1) % 3][0])
    torch.cuda.synchronize()
    print("%-12s %.3f ms per step" % (name, (time.perf_counter() - t0) / 30 * 1e3))
