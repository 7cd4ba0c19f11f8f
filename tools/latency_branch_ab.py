"""One batch in flight (configs[1], B=16, N=8192): the single-stream graph against the graph with the coordinate-only work
(pn2_coarse_geometry, FP4's three_nn) on a parallel branch (get_sa_fp_features(side_stream=)), and -- when the library has
them -- the split sampler variants.  Replay + synchronize per step, interleaved, min / median of 5 x 40.
    gpurun -- 'python tools/latency_branch_ab.py'"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pn2_amd as pn2  # noqa: E402
import bench  # noqa: E402
dev = torch.device("cuda:0")
hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
B, N = 16, 8192
hp.update(batch_size=B, num_point=N)
tfu = pn2.util.tf_util
tfu.set_default_store(tfu.VariableStore(device=dev, seed=1))
pc = torch.from_numpy(bench.s_scene(1, B, N)).to(dev)
side = torch.cuda.Stream()
variants = {
    "one stream": lambda x: pn2.model.get_sa_fp_features(x, False, hp)[0],
    "sa1 hand-over, one stream": lambda x: pn2.model.get_sa_fp_features(x, False, hp, sa1=pn2.model.sa1_samples(x, hp))[0],
    "branch: coarse + three_nn": lambda x: pn2.model.get_sa_fp_features(x, False, hp, sa1=pn2.model.sa1_samples(x, hp), side_stream=side)[0],
}
if hasattr(pn2.runtime, "latency_forward"):
    for chunks in ((768, 1024), (512, 768, 1024), (256, 512, 768, 1024)):
        variants["split sampler %s" % (chunks,)] = (lambda c: lambda x: pn2.runtime.latency_forward(x, hp, side, c))(chunks)
caps = {k: pn2.runtime.CapturedForward(f, pc) for k, f in variants.items()}
ref = caps["one stream"].replay().clone()
res = {k: [] for k in caps}
for rep in range(5):
    for k, c in caps.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            c.replay()
            torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / 40 * 1e3)
for k, c in caps.items():
    same = torch.equal(c.replay(), ref)
    torch.cuda.synchronize()
    print("%-40s min %.4f  median %.4f ms   equal to the one-stream graph: %s" % (k, min(res[k]), float(np.median(res[k])), same))
