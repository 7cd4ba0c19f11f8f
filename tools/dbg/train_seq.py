"""one eager training step (capture off) for a kernel-order trace: which library copies / fills sit between our kernels"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pn2_amd as pn2
from bench import s_scene
dev = torch.device("cuda:0")
hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
B, N = 16, 8192
rs = np.random.RandomState(100)
pc = torch.from_numpy(np.concatenate([s_scene(3000, B, N)[:, :, :3], rs.random_sample((B, N, 3)).astype(np.float32)], 2)).to(dev)
labels = torch.from_numpy(rs.randint(0, 9, (B, N)).astype(np.int64)).to(dev)
smpw = torch.from_numpy((rs.random_sample((B, N)) + 0.5).astype(np.float32)).to(dev)
tr = pn2.train.Trainer(hp, 9, store=pn2.util.tf_util.VariableStore(device=dev, seed=0), device=dev, capture=False)
for i in range(3):
    tr.train_step(pc, labels, smpw)
torch.cuda.synchronize()
marker = torch.zeros(7, device=dev); marker.fill_(1.0)   # a recognisable fill before the traced step
torch.cuda.synchronize()
tr.train_step(pc, labels, smpw)
torch.cuda.synchronize()
