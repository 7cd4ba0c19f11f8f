"""Where the time between "the captured training step replayed alone" and "a training step" goes (round 6: 3.35 vs 3.77 ms at first;
3.22 vs 3.45 after the staged hand-over and the LDS scatter-plan build).

    gpurun -- 'python tools/train_gap_probe.py > gpurun_out/train_gap.txt'

The bench's two resident batches hold the SAME scenes, so the geometry in the graph's static buffers stays valid whatever the
side stream does: the prefetch can be thinned out piece by piece without changing the step's work.
  graph        tr._graph.replay() back to back
  glue         train_step without any prefetch and without the geometry copy (fills, input copy, events, the replay)
  +copy        glue + the per-step copy of the 25 geometry tensors into the static buffers
  +fps         glue + SA1's sampler + ball query on the side stream (result discarded)
  +geometry    glue + the whole geometry chain without the scatter plans (discarded)
  +plans       glue + the chain with the 7 scatter plans (discarded)
  full         what bench.py --train times
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import pn2_amd as pn2  # noqa: E402
from bench import s_scene  # noqa: E402

dev = torch.device("cuda:0")
B, N, K = 16, 8192, 40
hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
rs = np.random.RandomState(100)
pc = torch.from_numpy(np.concatenate([s_scene(3000, B, N), rs.random_sample((B, N, 3)).astype(np.float32)], 2)).to(dev)
labels = torch.from_numpy(rs.randint(0, 9, (B, N)).astype(np.int64)).to(dev)
smpw = torch.from_numpy((rs.random_sample((B, N)) + 0.5).astype(np.float32)).to(dev)
tr = pn2.train.Trainer(hp, 9, store=pn2.util.tf_util.VariableStore(device=dev, seed=0), device=dev)
pcs = [pc, pc.clone()]
for i in range(6):
    tr.train_step(pcs[i % 2], labels, smpw, next_pc=pcs[(i + 1) % 2])
torch.cuda.synchronize()


def timed(fn, k=K):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


def graph_only():
    with torch.cuda.stream(tr._stream):
        tr._graph.replay()


cnt = [0]


def step(next_pc=True):
    i = cnt[0]
    cnt[0] += 1
    tr.train_step(pcs[i % 2], labels, smpw, next_pc=pcs[(i + 1) % 2] if next_pc else None, sync=False)


res = {"graph": timed(graph_only), "full": timed(step)}
orig_geo_for, orig_prefetch = tr._geometry_for, tr._prefetch
geo_tensors = pn2.model.geometry_tensors
side = tr._geo_stream
xyz = tr._xyz_of(pc)


def with_side(work, copy=False):
    def prefetch(next_pc, after_event, next_labels=None, next_smpw=None):
        tr._staged_tag = None
        if work is None:
            return
        side.wait_event(after_event)
        with torch.cuda.stream(side):
            work()
    tr._prefetch = prefetch
    tr._geometry_for = lambda p, caller: tr._static_geo
    pn2.model.geometry_tensors = geo_tensors if copy else (lambda g: [])
    try:
        return timed(step)
    finally:
        tr._prefetch, tr._geometry_for, pn2.model.geometry_tensors = orig_prefetch, orig_geo_for, geo_tensors


res["glue"] = with_side(None)
# the copy: static buffers copied onto a clone of themselves would alias -- copy FROM a second set instead
clone = pn2.model.clone_geometry(tr._static_geo, tr._static_geo["xyzs"][0])
tr_geo_for = lambda p, caller: clone  # noqa: E731
tr._prefetch = lambda *a: None
tr._geometry_for = tr_geo_for
res["+copy"] = timed(step)
tr._prefetch, tr._geometry_for = orig_prefetch, orig_geo_for
res["+fps"] = with_side(lambda: pn2.util.pointnet_util.sa_geometry(xyz, hp["l1_npoint"], hp["l1_radius"], hp["l1_nsample"]))
res["+geometry"] = with_side(lambda: pn2.model.compute_geometry(xyz, hp, plans=False))
res["+plans"] = with_side(lambda: pn2.model.compute_geometry(xyz, hp, plans=True))
res["full_again"] = timed(step)
for k, v in res.items():
    print("%-12s %.4f ms per step" % (k, v))
