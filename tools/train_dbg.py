import sys, os, numpy as np, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pn2_amd as pn2
import bench
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
B, N = 16, 8192
rs = np.random.RandomState(100)
pc = torch.from_numpy(np.concatenate([bench.s_scene(3000, B, N), rs.random_sample((B, N, 3)).astype(np.float32)], 2)).to(dev)
labels = torch.from_numpy(rs.randint(0, 9, (B, N)).astype(np.int64)).to(dev)
smpw = torch.from_numpy((rs.random_sample((B, N)) + 0.5).astype(np.float32)).to(dev)
mode = sys.argv[1]
if "nogemm" in mode: pn2.util.tf_util.USE_HIP_GEMM = False
tr = pn2.train.Trainer(hp, 9, store=pn2.util.tf_util.VariableStore(device=dev, seed=0), device=dev, capture=("eager" not in mode))
for i in range(5):
    l = tr.train_step(pc, labels, smpw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    l = tr.train_step(pc, labels, smpw, sync=("nosync" not in mode))
    if "print" in mode: print(i, float(l), flush=True)
torch.cuda.synchronize()
print(mode, "ok ms/step", (time.perf_counter() - t0) / 20 * 1e3, float(l))
