"""End-to-end inference on one MI355X, mirroring what the reference's Predictor does per batch (predict.py:15-118):
    get_model (SA x4 + FP x4 + head)  ->  argmax  ->  interpolate_label_with_color onto the dense cloud,
on synthetic scenes with random-init weights (no dataset / checkpoint is available offline).  Everything between the
input batch and the dense labels stays on the device; the forward is replayed from one hipGraph.

    python examples/predict_synthetic.py [--batch 16] [--points 8192] [--dense 4000000]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pn2_amd as pn2  # noqa: E402


def scene(rs, b, n):
    """10 m x 10 m column standing on z = 0 with rgb in [0, 1) (dataset/semantic_dataset.py:109-121, semantic.json)."""
    xy = rs.uniform(-5, 5, (b, n, 2))
    z = np.clip(np.abs(rs.normal(0, 1.5, (b, n, 1))), 0, 8)
    return np.concatenate([xy, z, rs.random_sample((b, n, 3))], axis=2).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--points", type=int, default=8192)
    ap.add_argument("--dense", type=int, default=4000000, help="dense points to label by 3-NN vote")
    ap.add_argument("--num-class", type=int, default=9)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(0)
    hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
    hp["batch_size"], hp["num_point"] = args.batch, args.points
    pn2.util.tf_util.set_default_store(pn2.util.tf_util.VariableStore(device=dev, seed=0))
    batch = torch.from_numpy(scene(rs, args.batch, args.points)).to(dev)

    def forward(pc):
        logits, _ = pn2.model.get_model(pc, False, args.num_class, hp)
        return logits.argmax(dim=2).to(torch.int32)

    predictor = pn2.runtime.CapturedForward(forward, batch)  # one hipGraph: ~40 kernels, no Python in the loop
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        pred = predictor(batch)
    torch.cuda.synchronize()
    t_fwd = (time.perf_counter() - t0) / 10
    # label the dense cloud from the sparse predictions (predict.py:86-103: batch flattened to one sparse set)
    sparse_points = batch[:, :, :3].reshape(-1, 3).contiguous()
    sparse_labels = pred.reshape(-1).contiguous()
    dense = torch.from_numpy(scene(rs, 1, args.dense)[0, :, :3]).to(dev)
    pn2.interpolate_label_with_color(sparse_points, sparse_labels, dense, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dense_labels, dense_colors = pn2.interpolate_label_with_color(sparse_points, sparse_labels, dense, 3)
    torch.cuda.synchronize()
    t_int = time.perf_counter() - t0
    hist = torch.bincount(dense_labels.clamp(min=0), minlength=args.num_class).tolist()
    print("forward + argmax: %.3f ms per %d x %d batch (%.1f M points/s)" % (t_fwd * 1e3, args.batch, args.points,
                                                                            args.batch * args.points / t_fwd * 1e-6))
    print("label interpolation: %d sparse -> %d dense points in %.2f ms (%.0f M points/s)" % (
        sparse_points.shape[0], args.dense, t_int * 1e3, args.dense / t_int * 1e-6))
    print("dense label histogram:", hist, " colours:", tuple(dense_colors.shape), dense_colors.dtype)
    return dense_labels, dense_colors


if __name__ == "__main__":
    main()
