#!/usr/bin/env python
"""The reference's training loop (train.py:333-465) end to end on the device, on a synthetic scene:

    scene resident in HBM  ->  SemanticFileData.sample_batch (device scene sampler, N4)
                           ->  Trainer.train_step (SA/FP stack + head + weighted CE + backward + Adam, N1),
                               the NEXT batch's FPS / ball-query / three_nn chain prefetched on a side stream

No batch ever crosses PCIe.  usage: python examples/train_synthetic.py [steps] [scene_points]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pn2_amd as pn2  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    n_scene = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(0)
    # a 60 m x 40 m scene: ground + a few "buildings"; label = height band (something learnable from xyz + rgb)
    xy = np.stack([rs.uniform(0, 60, n_scene), rs.uniform(0, 40, n_scene)], 1)
    z = np.abs(rs.normal(0, 1.0, n_scene)) + 4.0 * ((xy[:, 0] // 10 + xy[:, 1] // 10) % 3 == 0) * rs.uniform(0, 1, n_scene)
    points = np.concatenate([xy, z[:, None]], 1).astype(np.float32).astype(np.float64)
    labels = np.clip((z / 0.7).astype(np.int32) + 1, 1, 8)
    colors = np.clip(np.stack([z / 5.0, xy[:, 0] / 60.0, xy[:, 1] / 40.0], 1) + rs.normal(0, 0.05, (n_scene, 3)), 0, 1)
    fd = pn2.dataset.SemanticFileData(points=points, labels=labels, colors=colors, box_size_x=10, box_size_y=10, device=dev)
    counts = np.bincount(labels, minlength=9).astype(np.float32)
    label_weights = torch.from_numpy(1.0 / np.log(1.2 + counts / counts.sum())).float().to(dev)  # semantic_dataset.py:282-290

    hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
    B, N = hp["batch_size"], hp["num_point"]
    tr = pn2.train.Trainer(hp, 9, store=pn2.util.tf_util.VariableStore(device=dev, seed=0), device=dev)

    def batch():
        c, _, l, col = fd.sample_batch(B, N, capacity=400000)
        return torch.cat([c, col], dim=2), l.long(), label_weights[l.long()]

    prep = torch.cuda.Stream()  # batch k+1 is sampled (and its geometry chain run) while step k trains
    cur = batch()
    losses = []
    t0 = None
    for i in range(steps):
        loss = tr.train_step(*cur, sync=False)          # enqueue step k
        with torch.cuda.stream(prep):
            nxt = batch()                                # sample batch k+1 beside it
            tr.prefetch_geometry(nxt[0])                 # ... then its FPS / ball-query / three_nn chain
        torch.cuda.current_stream().wait_stream(prep)    # step k+1 (enqueued next) consumes it
        for t in nxt:
            t.record_stream(torch.cuda.current_stream())
        losses.append(loss.clone())                      # the captured step returns ONE static device scalar: keep a copy per step
        cur = nxt
        if i == 7:  # eager warm-up steps and the one-time graph capture are behind us
            torch.cuda.synchronize()
            t0, i0 = time.perf_counter(), i + 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / max(1, steps - i0)
    fd.check_last()
    ls = [float(x) for x in losses]
    print("loss: first %.3f  ->  last %.3f (min %.3f)" % (ls[0], ls[-1], min(ls)))
    print("%.2f ms per step incl. scene sampling (%d scenes x %d points sampled on the device from a %d-point scene): %.1f M points/s"
          % (dt * 1e3, B, N, n_scene, B * N / dt / 1e6))


if __name__ == "__main__":
    main()
