"""Generates tests/golden/*.npz.  Run from the repo root: python tests/golden/make_golden.py

Two kinds of vectors:
  * reference_three_nn.npz -- the ONLY forward known-answer the reference repo holds for this
    path: the expected print-out in tf_ops/test_interpolate.py:30-35 (9 squared distances, 9
    indices for np.random.seed(100) inputs).  The numbers below are transcribed from that file;
    they were not produced by our code.
  * oracle_*.npz -- outputs of OUR oracle (oracle/pn2_oracle.c) on small seeded inputs, frozen
    so that a later change to the oracle or to the HIP kernels cannot drift silently.  The
    reference itself (TensorFlow 1.x + CUDA + Open3D) cannot be imported or built here, so these
    are regression pins, not reference outputs ("parity unpinned" for FPS / ball query, see
    DESIGN.md).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
from conftest import s_grid, s_randn, s_scene  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    np.savez(os.path.join(HERE, "reference_three_nn.npz"),
             seed=100, target_shape=(64, 8192, 3), reference_shape=(64, 1024, 3),
             dist=np.array([0.00175864, 0.00671887, 0.0034472, 0.00337327, 0.00191902, 0.00075543,
                            0.00169418, 0.00473733, 0.00381071], dtype=np.float64),
             idx=np.array([137, 856, 116, 76, 915, 199, 117, 659, 786], dtype=np.int32),
             printed="[0.00175864 0.00671887 0.0034472  0.00337327 0.00191902 0.00075543\n"
                     " 0.00169418 0.00473733 0.00381071]")
    out = {}
    # config[0] of BASELINE.json: B=2, N=1024, npoint=256, K=16, C=3, r=0.2 on U(0,1)^3
    rs = np.random.RandomState(7)
    xyz = rs.random_sample((2, 1024, 3)).astype(np.float32)
    feat = rs.random_sample((2, 1024, 3)).astype(np.float32)
    for mode in (0, 1, 2):
        f = O.farthest_point_sample(256, xyz, mode)
        nx = O.gather_point(xyz, f)
        idx, cnt = O.query_ball_point(0.2, 16, xyz, nx, mode)
        out["cfg0_fps_m%d" % mode] = f
        out["cfg0_bq_idx_m%d" % mode] = idx
        out["cfg0_bq_cnt_m%d" % mode] = cnt
    # grid input (mode independent, many exact ties)
    g = s_grid(3, 2, 1500, 64)
    fg = O.farthest_point_sample(200, g)
    out["grid_fps"] = fg
    gi, gc = O.query_ball_point(0.25, 32, g, O.gather_point(g, fg))
    out["grid_bq_idx"], out["grid_bq_cnt"] = gi, gc
    # scene-like input, semantic.json SA1 parameters on a small cloud
    sc = s_scene(5, 1, 2048)
    fs = O.farthest_point_sample(256, sc)
    out["scene_fps"] = fs
    si, scnt = O.query_ball_point(0.5, 32, sc, O.gather_point(sc, fs))
    out["scene_bq_idx"], out["scene_bq_cnt"] = si, scnt
    # three_nn / three_interpolate
    d, i3 = O.three_nn(xyz, nx)
    out["cfg0_nn_dist"], out["cfg0_nn_idx"] = d, i3
    w = O.fp_weights(d)
    out["cfg0_interp"] = O.three_interpolate(O.gather_point(feat, f), i3, w)
    np.savez_compressed(os.path.join(HERE, "oracle_small.npz"), **out)
    print("wrote", sorted(out))
    more = extra_fixtures()
    np.savez_compressed(os.path.join(HERE, "oracle_extra.npz"), **more)
    print("wrote", sorted(more))


def extra_inputs():
    """Seeded inputs of the fixtures for the ops added after the first freeze (shared with the tests)."""
    rs = np.random.RandomState(11)
    return dict(
        dist=rs.random_sample((2, 5, 300)).astype(np.float32),                       # select_top_k
        kx1=rs.random_sample((2, 400, 3)).astype(np.float32), kx2=rs.random_sample((2, 30, 3)).astype(np.float32),
        sp=s_scene(8, 1, 3000)[0], sl=rs.randint(0, 9, 3000).astype(np.int32), dp=s_scene(9, 1, 9000)[0],
        pw=rs.random_sample((2, 9000)).astype(np.float32), pr=rs.random_sample((2, 500)).astype(np.float32),
        logits=rs.randn(2, 50, 9).astype(np.float32), labels=rs.randint(0, 9, (2, 50)),
        smpw=(rs.random_sample((2, 50)) * (rs.random_sample((2, 50)) > 0.2)).astype(np.float32),
        bf=rs.randn(1000).astype(np.float32) * np.float32(37.0))


def extra_fixtures():
    x = extra_inputs()
    out = {}
    out["topk_idx"], out["topk_val"] = O.select_top_k(7, x["dist"])
    out["knn_val"], out["knn_idx"] = O.knn_point(5, x["kx1"], x["kx2"])
    for k in (1, 3, 8):
        out["label_k%d" % k], out["color_k%d" % k] = O.interpolate_label_with_color(x["sp"], x["sl"], x["dp"], k)
    out["prob_idx"], out["prob_cumsum"] = O.prob_sample(x["pw"], x["pr"])
    out["ce"] = np.float64(O.weighted_sparse_ce(x["logits"], x["labels"], x["smpw"]))
    out["bf16"] = O.bf16_round(x["bf"])
    return out


if __name__ == "__main__":
    main()
