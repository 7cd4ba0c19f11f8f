#!/usr/bin/env python
"""bench.py -- points/sec through the PointNet++ SA+FP stack on MI355X.

Workload (BASELINE.json configs[1]): full SSG PointNet++ of the reference's semantic.json --
4 set-abstraction + 4 feature-propagation modules, B=16 scenes x N=8192 points, xyz+rgb, fp32,
inference forward (the mode the reference's own benchmark.py times), synthetic "S-scene" input
(10 m x 10 m column), random-init weights (xavier, non-trivial BN stats).
One step = one forward pass of the stack over one batch already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 = one process per GPU over RCCL.  Started under torch.distributed.run (RANK / WORLD_SIZE in the
environment, the driver's way) the process is one rank; started bare with --gpus N > 1 it launches
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same args>`
itself and exits with that job's status.  Either way the line is only printed by a job whose world size equals
--gpus: a box with fewer than N GPUs, or a WORLD_SIZE that disagrees with --gpus, is a non-zero exit, never an
"n_gpus: 1" line.  The path shards over the batch dimension with no data-path collective (every rank runs its own
16 scenes: weak scaling); the timed region is bracketed by barrier + synchronize and the max over ranks is taken.
`--dry-run` (tests/test_dist_cpu.py) runs launcher, rendezvous (gloo), barriers, max-over-ranks and the JSON line
WITHOUT the GPU workload: the N > 1 plumbing checked on a CPU-only box.

Prints ONE JSON line (rank 0).  `value` / `ms_per_step` are the THROUGHPUT regime: `config.batches_in_flight`
(default 6) independent B=16 batches in flight on `config.streams` (4) streams -- runtime.StaggeredPipeline: a batch is
two hipGraphs (SA1's sampling | everything else) on its ONE stream, two of the four streams keep one sampled batch
ahead of their dense work so that a region that starts with empty queues does not run its batches in lockstep
(`--stagger off`: one graph per batch on `--pipeline` streams, the r01-r04 execution).  A step is still one full batch
through the whole stack, submitted and completed inside the timed region.  The LATENCY regime (one batch in flight, what the reference's benchmark.py
times) is reported next to it under `regimes`.  Besides the contract keys the line carries
  regimes       {"throughput": {...}, "latency": {...}}: ms per step and points/s of both regimes, each timed
                over the same K steps with barrier + synchronize on both sides; "throughput_steady_state": the throughput
                regime over 10 x K steps (K = 20 is a 9 ms region that starts with empty queues) -- never `value`
  roofline      the limiter OF THE TIMED (throughput) REGIME: with several batches in flight the latency-bound FPS
                chains (16 CUs each) hide behind the other batches' dense work and the step is the sum of the
                chip-filling kernels, which are MFMA work: `roofline` = the dominant one of them (largest share of
                that sum; algorithmic flops per launch / its HIP-event duration), `roofline.aggregate` = all MFMA
                flops of a step / ms_per_step.  `traffic` = HBM bytes per launch of that kernel from the committed PMC pass
                (`traffic_source`; counters cannot share a run with the timed region: tools/gpu_round4.sh).
  latency_limiter  the limiter of the latency regime: FPS, reported in its own units (ns per dependent round,
                distance evaluations/s, share of the single-batch latency) instead of an HBM fraction
  kernels       per-kernel accounting (algorithmic bytes|flops, HIP-event duration) for every kernel of the step
  other_inputs  the same graphs on the reference benchmark's own input (S-randn, benchmark.py:16-18) and on a cloud with a
                quarter of its rows duplicated (S-dup25): both regimes, the times of the data-dependent kernels (FPS, ball
                query, three_nn) and which of them are more than 1.5 x slower than on S-scene
  per_rank_points_per_s   every rank's own rate (N > 1: a slow rank is visible on the line)
  other_configs driver-timed lines for BASELINE configs[2] (MSG module), configs[4] (large scene, bf16) and
                configs[3] at one GPU (training step)
  north_star    the two kernel-level targets of BASELINE.json measured at their own shape
                (ball_query+group_point and the fused grouped MLP at B=16,N=8192,M=1024,K=32,C=128)
  cpu_baseline  the CPU oracle (oracle/, OpenMP C + numpy fp32) on the same workload, rank 0, N=1
`--train` (BASELINE configs[3]) prints the training line instead; at N > 1 it decomposes the step: allreduce_early_ms /
allreduce_late_ms (each gradient bucket's all-reduce alone), exposed_comm_ms (what a real step waits for them), ms_per_step_no_comm
and scaling_efficiency (the same job with its collectives skipped), per_rank_ms_per_step.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TF = 157.3   # dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)


def s_scene(seed, b, n):
    rs = np.random.RandomState(seed)
    xy = rs.uniform(-5, 5, (b, n, 2))
    z = np.clip(np.abs(rs.normal(0, 1.5, (b, n, 1))), 0, 8)
    rgb = rs.uniform(0, 1, (b, n, 3))
    return np.concatenate([xy, z, rgb], axis=2).astype(np.float32)


def s_randn(seed, b, n):
    """S-randn: the reference benchmark's own input (benchmark.py:16-18: np.random.randn(batch, num_point, 6))"""
    return np.random.RandomState(seed).randn(b, n, 6).astype(np.float32)


def s_dup25(seed, b, n):
    """S-dup25: S-scene with a quarter of its rows duplicates of other rows (xyz and colour), shuffled -- how the reference
    fills a cloud shorter than num_points_per_sample (dataset/semantic_dataset.py:101-106: np.random.choice of its own rows)"""
    rs = np.random.RandomState(seed + 7919)
    x = s_scene(seed, b, n)
    nd = n // 4
    for i in range(b):
        x[i, n - nd:] = x[i, rs.randint(0, n - nd, nd)]
        x[i] = x[i][rs.permutation(n)]
    return x


FPS_KERNELS = ("farthest_point_sample", "fps_gather", "fps_nested")
# one workgroup per cloud, latency-bound like the samplers (they hide behind the other batches when several are in flight)
LATENCY_KERNELS = FPS_KERNELS + ("coarse_geometry", "ball_query_bin")

# HBM bytes per launch of the roofline kernel: PMC counters are a separate rocprofv3 pass (--pmc FETCH_SIZE / WRITE_SIZE cannot
# share a run with the timed region), so the line cites the committed summary of that pass (tools/gpu_round4.sh,
# tools/pmc_to_profiles.py: gfx950 corrections as MI355X_MICROARCH.md prescribes) instead of carrying `null`.
PMC_FILE = "profiles/r06_pmc_hbm_traffic.json"
PMC_KERNEL_OF = {"fp_mlp_fused_pre": "fp_chain_pipe_kernel", "fp_mlp_fused": "sa_fused_kernel<2", "sa_mlp_max_fused": "sa_fused_kernel<3, 1, 1, 2"}


def pmc_traffic(kernel):
    """bytes per launch of `kernel` from the committed PMC summary, or None (file absent / kernel not in it)"""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            ks = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    key = PMC_KERNEL_OF.get(kernel, kernel)
    hits = [v["traffic_bytes"] for k, v in ks.items() if key in k]
    return max(hits) if hits else None


# ---- algorithmic work per launch (compulsory traffic) -----------------
def kernel_model(name, a):
    """-> (bound, units) with units = algorithmic bytes (hbm) or flops (mfma) of ONE launch."""
    if name in ("pn2_farthest_point_sample", "pn2_fps_gather", "pn2_fps_nested"):
        b, n, m = a[0], a[1], a[2]
        return "hbm", b * n * 12 + b * m * 4 + (b * m * 12 if name != "pn2_farthest_point_sample" else 0)
    if name == "pn2_coarse_geometry":  # ints: b, n0, nlev, fps mode, bq mode, then (decoded by _lib) npoint[], nsample[]
        b, n, L = a[:3]
        byts = b * n * 12
        for m, ns in zip(a[5:5 + L], a[5 + L:5 + 2 * L]):
            byts += b * (m * 16 + m * ns * 4 + m * 4 + n * 24)
            n = m
        return "hbm", byts
    if name == "pn2_gather_point":
        b, n, m = a[:3]
        return "hbm", b * m * 4 + b * m * 12 * 2
    if name in ("pn2_query_ball_point", "pn2_query_ball_point_binned"):
        b, n, m, _, ns = a[:5]
        return "hbm", b * n * 12 + b * m * 12 + b * m * ns * 4 + b * m * 4
    if name == "pn2_ball_query_bin":
        b, n = a[:2]
        return "hbm", b * n * 12 + b * n * 14
    if name == "pn2_group_point":
        b, n, c, m, ns = a[:5]
        return "hbm", b * m * ns * 4 + b * n * c * 4 + b * m * ns * c * 4
    if name == "pn2_sa_group_concat":
        b, n, m, ns, c = a[:5]
        return "hbm", b * m * ns * 4 + b * n * (3 + c) * 4 + b * m * 12 + b * m * ns * (3 + c) * 4
    if name == "pn2_three_nn":
        b, n, m = a[:3]
        return "hbm", b * n * 12 + b * m * 12 + b * n * 24
    if name == "pn2_three_interpolate":
        b, m, c, n = a[:4]
        return "hbm", b * m * c * 4 + b * n * 24 + b * n * c * 4
    if name == "pn2_fp_interp_concat":
        b, n, m, c1, c2 = a[:5]
        return "hbm", b * m * c2 * 4 + b * n * 24 + b * n * c1 * 4 + b * n * (c1 + c2) * 4
    if name == "pn2_linear":
        rows, cin, cout = a[:3]
        return "mfma", 2 * rows * cin * cout
    if name == "pn2_mlp_chain":
        rows, cin, L = a[0], a[1], a[2]
        widths = a[4:4 + L]
        fl = 0
        for w in widths:
            fl += 2 * rows * cin * w
            cin = w
        return "mfma", fl
    if name == "pn2_mlp_wide":      # ints: rows, cin, x_stride, nlayers, relu_last, pool, widths...
        rows, cin, L = a[0], a[1], a[3]
        fl = 0
        for w in a[6:6 + L]:
            fl += 2 * rows * cin * w
            cin = w
        return "mfma", fl
    if name == "pn2_sa_mlp_wide":   # ints: b, n, m, nsample, c, nlayers, pool, widths...
        b, n, m, ns, c, L = a[:6]
        cin, fl = 3 + c, 0
        for w in a[7:7 + L]:
            fl += 2 * b * m * ns * cin * w
            cin = w
        return "mfma", fl
    if name == "pn2_fp_mlp_wide":   # ints: b, n, m, c1, c2, nlayers, widths...
        b, n, m, c1, c2, L = a[:6]
        cin, fl = c1 + c2, 0
        for w in a[6:6 + L]:
            fl += 2 * b * n * cin * w
            cin = w
        return "mfma", fl
    if name == "pn2_fp_mlp_fused":
        b, n, m, c1, c2, L = a[:6]
        widths = a[6:6 + L]
        cin, fl = c1 + c2, 0
        for w in widths:
            fl += 2 * b * n * cin * w
            cin = w
        return "mfma", fl
    if name == "pn2_fp_mlp_wide_pre":  # ints: b, n, m, c1, nlayers, widths...   EXECUTED flops
        b, n, m, c1, L = a[:5]
        widths = a[5:5 + L]
        fl, cin = 2 * b * n * c1 * widths[0], widths[0]
        for w in widths[1:]:
            fl += 2 * b * n * cin * w
            cin = w
        return "mfma", fl
    if name == "pn2_sa_mlp_wide_pre":  # ints: b, n, m, nsample, nlayers, pool, widths...
        b, n, m, ns, L = a[:5]
        widths = a[6:6 + L]
        fl, cin = 2 * b * m * ns * 3 * widths[0], widths[0]
        for w in widths[1:]:
            fl += 2 * b * m * ns * cin * w
            cin = w
        return "mfma", fl
    if name == "pn2_sa_mlp_fused_pre":  # ints: b, n, m, nsample, nlayers, pool, widths...  EXECUTED flops (xyz rows + later layers)
        b, n, m, ns, L = a[:5]
        widths = a[6:6 + L]
        fl = 2 * b * m * ns * 3 * widths[0]
        cin = widths[0]
        for w in widths[1:]:
            fl += 2 * b * m * ns * cin * w
            cin = w
        return "mfma", fl
    if name == "pn2_fp_mlp_fused_pre":  # ints: b, n, m, c1, nlayers, widths...  EXECUTED flops: skip channels + later layers
        b, n, m, c1, L = a[:5]
        widths = a[5:5 + L]
        fl = 2 * b * n * c1 * widths[0]
        cin = widths[0]
        for w in widths[1:]:
            fl += 2 * b * n * cin * w
            cin = w
        return "mfma", fl
    if name in ("pn2_sa_mlp_max_fused", "pn2_sa_mlp_rows_fused"):
        b, n, m, ns, c, L = a[:6]
        widths = a[6:6 + L]
        cin, fl = 3 + c, 0
        for w in widths:
            fl += 2 * b * m * ns * cin * w
            cin = w
        return "mfma", fl
    return "hbm", 0


def reference_flops(pn2, hp, B, N):
    """MFMA flops of one step in the reference's own formulation (every 1x1 conv of every SA / FP module applied to the
    grouped / concatenated tensor: SURVEY.md 8(d)'s 2*rows*cin*cout).  The product executes fewer: the first layer of a
    module is applied to the SOURCE rows where linearity allows (DESIGN.md 4 'hoisting')."""
    mdl = pn2.model
    npts = [N] + [hp["l%d_npoint" % i] for i in (1, 2, 3, 4)]
    width = [3 * int(hp["use_color"])] + [w[-1] for w in mdl.SA_MLPS]
    fl = 0
    for li in range(4):
        rows, cin = B * npts[li + 1] * hp["l%d_nsample" % (li + 1)], 3 + width[li]
        for w in mdl.SA_MLPS[li]:
            fl, cin = fl + 2 * rows * cin * w, w
    up = width[4]
    for fi in range(4):
        lvl = 3 - fi
        rows, cin = B * npts[lvl], width[lvl] + up
        for w in mdl.FP_MLPS[fi]:
            fl, cin = fl + 2 * rows * cin * w, w
        up = cin
    return fl


def summarize_trace(trace, steps):
    """aggregate (name, args) -> avg ms per launch, launches per step, roofline numbers."""
    agg = {}
    for name, args, s, e in trace:
        key = (name, args)
        d = agg.setdefault(key, [0.0, 0])
        d[0] += s.elapsed_time(e)
        d[1] += 1
    rows = []
    for (name, args), (tot_ms, cnt) in agg.items():
        bound, units = kernel_model(name, args)
        avg_ms = tot_ms / cnt
        if bound == "hbm":
            ach = units / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            peak, unit = HBM_PEAK_GBS, "GB/s"
        else:
            ach = units / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
            peak, unit = MFMA_F32_PEAK_TF, "TFLOP/s"
        rows.append({"kernel": name.replace("pn2_", ""), "args": list(args), "bound": bound,
                     "avg_us": round(avg_ms * 1e3, 2), "launches_per_step": cnt / steps,
                     "ms_per_step": round(tot_ms / steps, 4), "achieved": round(ach, 3), "peak": peak,
                     "unit": unit, "frac": round(ach / peak, 5),
                     "algorithmic_units": int(units)})
    rows.sort(key=lambda r: -r["ms_per_step"])
    return rows


def time_call(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters  # ms


def north_star_kernels(pn2, dev):
    """ball_query+group_point and fused grouped MLP at B=16,N=8192,M=1024,K=32,C=128."""
    B, N, M, K, C = 16, 8192, 1024, 32, 128
    pc = s_scene(0, B, N)
    xyz = torch.from_numpy(pc[:, :, :3].copy()).to(dev)
    feat = torch.from_numpy(np.random.RandomState(1).randn(B, N, C).astype(np.float32)).to(dev)
    new_xyz = pn2.gather_point(xyz, pn2.farthest_point_sample(M, xyz))
    idx, _ = pn2.query_ball_point(0.5, K, xyz, new_xyz)
    g = pn2.tf_ops.tf_grouping
    bins = g.ball_query_bin_alloc(xyz)
    t_bin = time_call(lambda: g.ball_query_bin(0.5, xyz, out=bins), 20)
    idx_b, _ = g.query_ball_point_binned(0.5, K, xyz, new_xyz, bins)
    assert torch.equal(idx_b, idx)
    t_bq_self = time_call(lambda: pn2.query_ball_point(0.5, K, xyz, new_xyz), 20)           # every workgroup bins the cloud itself
    t_bq = time_call(lambda: g.query_ball_point_binned(0.5, K, xyz, new_xyz, bins), 20)     # variant: cloud binned once
    t_gp = time_call(lambda: pn2.group_point(feat, idx), 20)
    bq_bytes = B * N * 12 + B * M * 12 + B * M * K * 4 + B * M * 4
    gp_bytes = B * M * K * 4 + B * N * C * 4 + B * M * K * C * 4
    gbs = lambda t: (bq_bytes + gp_bytes) / (t * 1e-3) / 1e9  # noqa: E731
    ach = gbs(t_bq_self + t_gp)                                # the default product path: pn2_query_ball_point + pn2_group_point
    out = {"ball_query_group_point": {
        "shape": "B16 N8192 M1024 K32 C128", "ball_query_us": round(t_bq_self * 1e3, 1),
        "group_point_us": round(t_gp * 1e3, 1), "bytes": bq_bytes + gp_bytes, "bound": "hbm",
        "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
        "group_point_alone_GBs": round(gp_bytes / (t_gp * 1e-3) / 1e9, 1),
        "binned_once_variant": {
            "note": "pn2_ball_query_bin once per cloud + pn2_query_ball_point_binned (same indices): what the pipelined "
                    "(throughput) execution runs since r06 -- the binning rides in the sampler half of a batch (model.sa1_samples); "
                    "the one-batch graph keeps the self-binning query",
            "bin_us": round(t_bin * 1e3, 1), "ball_query_us": round(t_bq * 1e3, 1),
            "frac_query_only": round(gbs(t_bq + t_gp) / HBM_PEAK_GBS, 4),
            "frac_including_bin": round(gbs(t_bin + t_bq + t_gp) / HBM_PEAK_GBS, 4)}}}
    # the achievable copy bandwidth next to the 8 TB/s peak (device-to-device copy of the same 268 MB)
    grouped = pn2.group_point(feat, idx)
    dst = torch.empty_like(grouped)
    t_cp = time_call(lambda: dst.copy_(grouped), 20)
    cp = 2 * grouped.numel() * 4 / (t_cp * 1e-3) / 1e9
    out["ball_query_group_point"]["d2d_copy_probe_GBs"] = round(cp, 1)
    out["ball_query_group_point"]["frac_of_copy_probe"] = round(ach / cp, 4)
    del grouped, dst
    # fused grouped MLP: one 128 -> 128 layer (+3 xyz channels of the SA concat) + max over K
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    keep = tfu.get_default_store()
    tfu.set_default_store(tfu.VariableStore(device=dev, seed=2))
    try:
        with tfu.variable_scope("ns"):
            fn = lambda: pu._sa_fused_inference(xyz, new_xyz, feat, idx, [128], True, "conv%d")  # noqa: E731
            assert fn() is not None
            t_mlp = time_call(fn, 10)
    finally:
        tfu.set_default_store(keep)
    flops = 2 * B * M * K * (3 + C) * 128
    ach = flops / (t_mlp * 1e-3) / 1e12
    out["fused_grouped_mlp"] = {"shape": "B16 N8192 M1024 K32 Cin131 Cout128 + max", "us": round(t_mlp * 1e3, 1),
                                "flops": flops, "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TF,
                                "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TF, 4)}
    return out


def both_regimes(pn2, sampler_fn, dense_fn, make_batch, units_per_step, steps, regions=3):
    """a workload that starts with a sampler chain, timed like the headline -- `latency`: ONE batch in flight (one
    graph: sampler + rest, replay + synchronise per step) and `throughput`: runtime.StaggeredPipeline (two graphs per batch on the
    batch's one stream, 4 streams, backlogs 0,0,1,1), median of `regions` regions of `steps` steps after one untimed region.
    sampler_fn(x) -> s, dense_fn(x, s) -> y, make_batch(n) -> input of slot n."""
    cap = pn2.runtime.CapturedForward(lambda x: dense_fn(x, sampler_fn(x)), make_batch(0))
    for _ in range(2):
        cap.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        cap.replay()
        torch.cuda.synchronize()
    lat = (time.perf_counter() - t0) / steps * 1e3
    del cap
    pipe = pn2.runtime.StaggeredPipeline(sampler_fn, dense_fn, make_batch, (0, 0, 1, 1))

    def region():
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            pipe.step()
        pipe.flush()
        torch.cuda.synchronize()
        return (time.perf_counter() - t1) / steps * 1e3
    region()
    ts = sorted(region() for _ in range(regions))
    thr = ts[(len(ts) - 1) // 2]
    res = {"latency": {"batches_in_flight": 1, "ms_per_step": round(lat, 4), "points_per_s": round(units_per_step / (lat * 1e-3), 1)},
           "throughput": {"batches_in_flight": pipe.batches_in_flight, "streams": pipe.P,
                          "streams_verified_concurrent": pipe.streams_verified_concurrent, "ms_per_step": round(thr, 4),
                          "ms_per_step_regions": [round(v, 4) for v in ts], "points_per_s": round(units_per_step / (thr * 1e-3), 1)}}
    del pipe
    return res


def other_configs(pn2, dev, hp, steps):
    """configs[2], configs[4] and configs[3]@1GPU timed in the same run (graph replay where the path is captured)."""
    out = {}
    tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
    keep = tfu.get_default_store()
    try:
        # configs[2]: MSG set abstraction, 3 scales (radii / K / MLPs are builder-chosen: the reference ships none)
        B, N, M = 16, 8192, 1024
        radii, ks, mlps = [0.25, 0.5, 1.0], [16, 32, 64], [[32, 32, 64], [64, 64, 128], [64, 96, 128]]
        pc = torch.from_numpy(s_scene(5000, B, N)).to(dev)
        xyz, pts = pc[:, :, :3].contiguous(), pc[:, :, 3:].contiguous()
        tfu.set_default_store(tfu.VariableStore(device=dev, seed=1))
        cap = pn2.runtime.CapturedForward(
            lambda x: pu.pointnet_sa_module_msg(x, pts, M, radii, ks, mlps, False, None, scope="msg")[1], xyz)
        t = time_call(cap.replay, steps)
        out["configs[2]"] = {"workload": "MSG SA module, 3 scales r=(0.25,0.5,1.0) K=(16,32,64) "
                                         "MLPs ([32,32,64],[64,64,128],[64,96,128]) (builder-chosen), B=16, N=8192, npoint=1024, fp32",
                             "ms_per_step": round(t, 4), "points_per_s": round(B * N / (t * 1e-3), 1), "steps": steps,
                             "launch": "one hipGraph replay per step, one batch in flight"}
        del cap
        # both regimes (sampler | the rest of the module on the sampled centres), as the headline config gets them
        fps_gather = pn2.tf_ops.tf_sampling.farthest_point_sample_and_gather
        out["configs[2]"]["regimes"] = both_regimes(
            pn2, lambda x: fps_gather(M, x)[1],
            lambda x, nx: pu.pointnet_sa_module_msg(x, pts, M, radii, ks, mlps, False, None, scope="msg", new_xyz=nx)[1],
            lambda n: (xyz if n == 0 else torch.from_numpy(s_scene(5100 + n, B, N)[:, :, :3].copy()).to(dev)), B * N, steps)
        # configs[4]: large scenes, N=65536 -> npoint 4096, K=64, C=128 bf16 features, fused bf16 grouped MLP; B=1 (eager and
        # as a hipGraph replay) and B=16 ("B=1 (and 16 if memory allows)")
        N4, M4, K4, C4 = 65536, 4096, 64, 128
        tfu.set_default_store(tfu.VariableStore(device=dev, seed=2))
        res4 = {}
        for B4 in (1, 16):
            xyz4 = torch.from_numpy(s_scene(5001, B4, N4)[:, :, :3].copy()).to(dev)
            pts4 = torch.randn(B4, N4, C4, device=dev).to(torch.bfloat16)

            def sa4(x):
                _, nx = pn2.tf_ops.tf_sampling.farthest_point_sample_and_gather(M4, x)
                idx, _ = pn2.query_ball_point(0.5, K4, x, nx)
                with tfu.variable_scope("sa"):
                    return pu.sa_features_inference(x, nx, pts4, idx, [128, 128])
            with torch.no_grad():
                sa4(xyz4)
                n4 = max(2, min(steps, 5))
                t_eager = time_call(lambda: sa4(xyz4), n4, warmup=1)
                t_fps = time_call(lambda: pn2.tf_ops.tf_sampling.farthest_point_sample_and_gather(M4, xyz4), n4, warmup=1)
            cap4 = pn2.runtime.CapturedForward(sa4, xyz4)
            t_graph = time_call(cap4.replay, n4, warmup=1)
            reg4 = None
            if B4 == 16:  # both regimes on the B = 16 form (sampler | ball query + fused bf16 MLP)

                def dense4(x, nx):
                    idx, _ = pn2.query_ball_point(0.5, K4, x, nx)
                    with tfu.variable_scope("sa"):
                        return pu.sa_features_inference(x, nx, pts4, idx, [128, 128])
                with torch.no_grad():
                    reg4 = both_regimes(pn2, lambda x: pn2.tf_ops.tf_sampling.farthest_point_sample_and_gather(M4, x)[1], dense4,
                                        lambda n: (xyz4 if n == 0 else
                                                   torch.from_numpy(s_scene(5200 + n, B4, N4)[:, :, :3].copy()).to(dev)),
                                        B4 * N4, n4)
            res4[B4] = (t_eager, t_graph, t_fps, n4, reg4)
            del cap4, xyz4, pts4
        t_eager, t_graph, t_fps, n4, _ = res4[1]
        out["configs[4]"] = {"workload": "large-scene SA layer: B=1, N=65536, npoint=4096, K=64, C=128 bf16 features, "
                                         "FPS (lazy multi-pick over Hilbert-sorted buckets) + ball query + fused bf16 grouped MLP [128,128] + max",
                             "ms_per_step": round(t_graph, 4), "points_per_s": round(N4 / (t_graph * 1e-3), 1), "steps": n4,
                             "fps_ms": round(t_fps, 4), "launch": "one hipGraph replay per step", "eager_ms_per_step": round(t_eager, 4),
                             "B16": {"ms_per_step": round(res4[16][1], 4), "points_per_s": round(16 * N4 / (res4[16][1] * 1e-3), 1),
                                     "fps_ms": round(res4[16][2], 4), "eager_ms_per_step": round(res4[16][0], 4),
                                     "launch": "one hipGraph replay per step, 16 scenes per step", "regimes": res4[16][4]}}
    except Exception as ex:  # keep the headline line alive
        out["error"] = repr(ex)
    finally:
        tfu.set_default_store(keep)
    try:
        # configs[3] on this one GPU: a full training step (forward with batch-stat BN, weighted CE, backward, Adam)
        B, N = hp["batch_size"], hp["num_point"]
        rs = np.random.RandomState(100)
        pc = torch.from_numpy(np.concatenate([s_scene(3000, B, N)[:, :, :3], rs.random_sample((B, N, 3)).astype(np.float32)], 2)).to(dev)
        labels = torch.from_numpy(rs.randint(0, 9, (B, N)).astype(np.int64)).to(dev)
        smpw = torch.from_numpy((rs.random_sample((B, N)) + 0.5).astype(np.float32)).to(dev)
        tr = pn2.train.Trainer(hp, 9, store=tfu.VariableStore(device=dev, seed=0), device=dev)
        pcs = [pc, pc.clone()]  # two resident batches, alternated (the trainer prefetches the next batch's geometry)
        w3 = tr.warmup_eager + 2
        for i in range(w3):  # eager steps, then the capture, then one replay: all outside the timing
            tr.train_step(pcs[i % 2], labels, smpw, next_pc=pcs[(i + 1) % 2], next_labels=labels, next_smpw=smpw)
        torch.cuda.synchronize()
        n3 = max(2, min(steps, 10))
        t0 = time.perf_counter()
        for i in range(w3, w3 + n3):
            loss = tr.train_step(pcs[i % 2], labels, smpw, next_pc=pcs[(i + 1) % 2], next_labels=labels, next_smpw=smpw, sync=False)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / n3 * 1e3
        out["configs[3]@1gpu"] = {"workload": "training step (forward with batch-stat BN + weighted CE + backward + Adam), "
                                              "%d scenes x %d points, fp32; the multi-GPU line is `bench.py --train --gpus N`" % (B, N),
                                  "ms_per_step": round(t, 4), "points_per_s": round(B * N / (t * 1e-3), 1), "steps": n3,
                                  "last_loss": float(loss)}
    except Exception as ex:
        out["configs[3]@1gpu"] = {"error": repr(ex)}
    finally:
        tfu.set_default_store(keep)
    return out


def cpu_baseline(pn2, store, pc, hp):
    """The CPU oracle (OpenMP C restatement of the reference kernels + numpy fp32 for the dense
    layers) on the same workload and weights, on this box's host cores.  Bounded: one pass over
    the full batch after a 1-scene warm-up."""
    from oracle import oracle as O

    def layer_dicts(scope, names):
        out = []
        for nm in names:
            p = "%s/%s/" % (scope, nm)
            W = store.params[p + "weights"].detach().cpu().numpy()
            out.append(dict(W=W.reshape(W.shape[-2], W.shape[-1]), b=store.params[p + "biases"].detach().cpu().numpy(),
                            gamma=store.params[p + "bn/gamma"].detach().cpu().numpy(),
                            beta=store.params[p + "bn/beta"].detach().cpu().numpy(),
                            mean=store.buffers[p + "bn/moving_mean"].cpu().numpy(),
                            var=store.buffers[p + "bn/moving_variance"].cpu().numpy()))
        return out

    sa_layers = [layer_dicts("layer%d" % (i + 1), ["conv0", "conv1", "conv2"]) for i in range(4)]
    fp_layers = [layer_dicts("fa_layer%d" % (i + 1), ["conv_%d" % j for j in range(len(pn2.model.FP_MLPS[i]))])
                 for i in range(4)]

    def run(x):
        xyzs, feats = [x[:, :, :3]], [x[:, :, 3:6]]
        for li in range(4):
            k = "l%d_" % (li + 1)
            nx, npts, _ = O.sa_module(xyzs[-1], feats[-1], hp[k + "npoint"], hp[k + "radius"], hp[k + "nsample"],
                                      sa_layers[li], dtype=np.float32)
            xyzs.append(nx)
            feats.append(npts)
        up = feats[4]
        for fi in range(4):
            lvl = 3 - fi
            up = O.fp_module(xyzs[lvl], xyzs[lvl + 1], feats[lvl], up, fp_layers[fi], dtype=np.float32)
        return up

    run(pc[:1])
    t0 = time.time()
    passes = 0
    while passes < 20 and (passes == 0 or time.time() - t0 < 10.0):  # bounded: ~10 s of CPU work
        run(pc)
        passes += 1
    dt = (time.time() - t0) / passes
    return {"value": round(pc.shape[0] * pc.shape[1] / dt, 1), "unit": "points/s", "cores": O.num_threads(),
            "kind": "port", "seconds_per_pass": round(dt, 3), "passes": passes,
            "sample": "%d forward passes of the same SA+FP stack over the full batch (%d scenes x %d points): "
                      "OpenMP C oracle for FPS/ball query/group/three_nn/interpolate + numpy(BLAS) fp32 dense layers"
                      % (passes, pc.shape[0], pc.shape[1])}


def bench_train(pn2, args, hp, B, N, rank, world, dev):
    """configs[3]: every rank trains on its own 16 scenes; the only collective is the flat gradient all-reduce."""
    rs = np.random.RandomState(100 + rank)
    pc = torch.from_numpy(np.concatenate([s_scene(3000 + rank, B, N), rs.random_sample((B, N, 3)).astype(np.float32)], 2)).to(dev)
    labels = torch.from_numpy(rs.randint(0, 9, (B, N)).astype(np.int64)).to(dev)
    smpw = torch.from_numpy((rs.random_sample((B, N)) + 0.5).astype(np.float32)).to(dev)
    tr = pn2.train.Trainer(hp, 9, store=pn2.util.tf_util.VariableStore(device=dev, seed=0), device=dev)
    pcs = [pc, pc.clone()]  # two resident batches, alternated: the trainer prefetches the geometry of the next one
    for i in range(max(tr.warmup_eager + 2, args.warmup)):  # includes the one-time hipGraph capture of the step (1 GPU)
        tr.train_step(pcs[i % 2], labels, smpw, next_pc=pcs[(i + 1) % 2], next_labels=labels, next_smpw=smpw)
    pn2.dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    i0 = max(tr.warmup_eager + 2, args.warmup)
    for i in range(i0, i0 + args.steps):  # no host synchronisation inside the timed region: the loss stays on the device
        loss = tr.train_step(pcs[i % 2], labels, smpw, next_pc=pcs[(i + 1) % 2], next_labels=labels, next_smpw=smpw, sync=False)
    torch.cuda.synchronize()
    local = time.perf_counter() - t0  # this rank's own clock, before the closing barrier
    pn2.dist.barrier()
    elapsed = pn2.dist.max_over_ranks(time.perf_counter() - t0, device=dev)
    per_rank = pn2.dist.gather_over_ranks(local / args.steps * 1e3, device=dev)
    # ---- diagnosis legs, AFTER the measured region (they perturb the replicas): where a step's time goes when N > 1
    diag = train_comm_diagnosis(pn2, tr, args, pcs, labels, smpw, dev, world, elapsed / args.steps * 1e3)
    # what the concurrent geometry chain of the NEXT batch costs the step: the captured step replayed back to back on the inputs
    # resident in its static buffers, nothing on the side stream (single rank; after the measured region: the weights move on)
    if world == 1 and tr._graph is not None and tr._graph_adam is None:
        with torch.cuda.stream(tr._stream):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                tr._graph.replay()
            torch.cuda.synchronize()
        diag["graph_replay_alone_ms"] = round((time.perf_counter() - t1) / args.steps * 1e3, 4)
    if rank == 0:
        print(json.dumps({
            "metric": "training points/sec through SA+FP stack + head (B=16/GPU, N=%d)" % N,
            "value": round(world * B * N * args.steps / elapsed, 1), "unit": "points/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[3]: data-parallel training, %d scenes x %d points per GPU, semantic.json, "
                                   "batch-stat BN, weighted CE, Adam, two-bucket gradient all-reduce (%d parameters), next batch's FPS/ball-query/three_nn "
                                   "chain prefetched on a side stream"
                                   % (B, N, tr.store.num_parameters()),
                       "global_batch": world * B, "parallelism": "dp%d" % world},
            "per_rank_ms_per_step": [round(v, 4) for v in per_rank], **diag,
            # (the training step has no stream pipeline to verify: one trainer stream + one geometry-prefetch stream per rank)
            "streams_verified_concurrent": [None] * world,
            "last_loss": float(loss), **rccl_info(world)}))


def train_comm_diagnosis(pn2, tr, args, pcs, labels, smpw, dev, world, ms_per_step):
    """the keys that let a sub-linear N > 1 result be read from the record.
      allreduce_early_ms / allreduce_late_ms   each bucket's all-reduce alone (nothing to hide behind): the wire
      exposed_comm_ms                          in a real step: end of the SA backward graph -> both buckets reduced (HIP
                                               events on the trainer's stream); what the step waits for the collectives
      ms_per_step_no_comm, scaling_efficiency  the SAME job stepping with its collectives skipped (max over ranks), and
                                               that over the measured step: 1.0 = the collectives cost nothing"""
    k = max(4, min(args.steps, 10))
    out = {"allreduce_early_ms": 0.0, "allreduce_late_ms": 0.0, "exposed_comm_ms": 0.0,
           "ms_per_step_no_comm": round(ms_per_step, 4), "scaling_efficiency": 1.0}
    if world == 1:
        return out
    i0 = tr.step_count
    tr.comm_events = []
    for i in range(i0, i0 + k):
        tr.train_step(pcs[i % 2], labels, smpw, next_pc=pcs[(i + 1) % 2], next_labels=labels, next_smpw=smpw, sync=False)
    torch.cuda.synchronize()
    ev, tr.comm_events = tr.comm_events, None
    if ev:
        out["exposed_comm_ms"] = round(sum(e1.elapsed_time(e2) for _, e1, e2 in ev) / len(ev), 4)
        out["early_launch_to_reduced_ms"] = round(sum(e0.elapsed_time(e2) for e0, _, e2 in ev) / len(ev), 4)
    out["exposed_comm_ms"] = round(pn2.dist.max_over_ranks(out["exposed_comm_ms"], device=dev), 4)
    # every key of this dict is a maximum over ranks (r06: this one was rank 0's own value beside the maximum above)
    out["early_launch_to_reduced_ms"] = round(pn2.dist.max_over_ranks(out.get("early_launch_to_reduced_ms", 0.0), device=dev), 4)
    pn2.dist.barrier()
    tr.bucket.skip_collectives = True
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(i0 + k, i0 + 2 * k):
        tr.train_step(pcs[i % 2], labels, smpw, next_pc=pcs[(i + 1) % 2], next_labels=labels, next_smpw=smpw, sync=False)
    torch.cuda.synchronize()
    no_comm = pn2.dist.max_over_ranks((time.perf_counter() - t0) / k * 1e3, device=dev)
    tr.bucket.skip_collectives = False
    out["ms_per_step_no_comm"] = round(no_comm, 4)
    out["scaling_efficiency"] = round(no_comm / ms_per_step, 4)
    tc = tr.bucket.time_collectives(iters=k)
    out.update({kk: (round(pn2.dist.max_over_ranks(v, device=dev), 4) if kk.endswith("_ms") else v) for kk, v in tc.items()})
    return out


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(args):
    """`bench.py --gpus N` started bare (no RANK in the environment): become the launcher of N ranks on this node."""
    import subprocess
    if not args.dry_run:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit("bench.py --gpus %d: this node has %d visible GPU(s); refusing to print a line for fewer "
                             "ranks than requested" % (args.gpus, have))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes on this driver)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def rccl_info(world):
    """what the collective layer really is: world size of the initialised group, backend, RCCL version."""
    import torch.distributed as dist
    info = {"rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1,
            "backend": dist.get_backend() if dist.is_initialized() else None}
    try:
        info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        info["rccl_version"] = None
    assert info["rccl_ranks"] == world, info
    return info


def dry_run(args, rank, world):
    """The N > 1 plumbing without the GPU workload (gloo on CPU): rendezvous, barrier-bracketed timed region,
    max over ranks, ONE line from rank 0."""
    import torch.distributed as dist
    import pn2_amd as pn2
    pn2.dist.init_from_env(backend="gloo")
    pn2.dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))  # rank-dependent "work": the slowest rank must define the step
    local = time.perf_counter() - t0
    pn2.dist.barrier()
    elapsed = pn2.dist.max_over_ranks(time.perf_counter() - t0)
    per_rank = pn2.dist.gather_over_ranks(local / args.steps * 1e3)
    line = {"metric": "dry run (no GPU work): launcher / rendezvous / max-over-ranks only", "value": None,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "dry_run": True,
            "per_rank_ms_per_step": [round(v, 4) for v in per_rank],
            "per_rank_points_per_s": [round(args.batch * args.points / (v * 1e-3), 1) for v in per_rank]}
    if args.train:
        # the diagnosis keys of `--train --gpus N` on the same two-bucket exchange (gloo, CPU tensors of the real sizes:
        # 967945 gradients, the head + FP layers in the early bucket)
        params = [torch.nn.Parameter(torch.zeros(n_)) for n_ in (300000, 667945)]
        bucket = pn2.dist.OverlappedGradAllReduce(params, 1)
        tc = bucket.time_collectives(iters=3)
        t1 = time.perf_counter()
        work = bucket.reduce_early_async()
        time.sleep(0.002)  # "the SA backward graph"
        t2 = time.perf_counter()
        bucket.reduce_late_and_wait(work)
        exposed = (time.perf_counter() - t2) * 1e3
        bucket.skip_collectives = True
        assert bucket.reduce_early_async() is None and bucket.world() == 1
        bucket.skip_collectives = False
        no_comm = line["ms_per_step"]
        line.update({k: (round(pn2.dist.max_over_ranks(v), 4) if k.endswith("_ms") else v) for k, v in tc.items()})
        line.update({"exposed_comm_ms": round(pn2.dist.max_over_ranks(exposed), 4),
                     "early_launch_to_reduced_ms": round((time.perf_counter() - t1) * 1e3, 4),
                     "ms_per_step_no_comm": no_comm,
                     "scaling_efficiency": round(no_comm / (no_comm + pn2.dist.max_over_ranks(exposed)), 4)})
    if rank == 0:
        line.update(rccl_info(world))
        print(json.dumps(line))
    pn2.dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / timing plumbing only, gloo on CPU, no GPU work (tests/test_dist_cpu.py)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--regions", type=int, default=5,
                    help="timed regions of exactly --steps steps each; `value` / `ms_per_step` are the median region's")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--points", type=int, default=8192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch from Python instead of replaying a hipGraph")
    ap.add_argument("--one-stream", action="store_true", help="(default since r01q; kept for old scripts)")
    ap.add_argument("--pipeline", type=int, default=None,
                    help="one graph per batch on this many streams (the r01-r04 execution; implies --stagger off)")
    ap.add_argument("--no-north-star", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the configs[2] / [3]@1gpu / [4] sub-results")
    ap.add_argument("--debug-set", action="append", default=[], metavar="WHAT=VALUE",
                    help="tuning hook: pn2_debug_set(what, value) before the run (A/B experiments)")
    ap.add_argument("--dup", action="append", default=[], metavar="ENTRY",
                    help="ablation: launch this entry point twice; the ms/step increase is its marginal cost in the "
                         "pipelined regime")
    ap.add_argument("--fp-front", choices=("auto", "fused", "unfused"), default="auto",
                    help="FP front end inside the first FP4 MLP kernel (pn2_fp_mlp_fused) or materialised by "
                         "pn2_fp_interp_concat; auto = fused (faster at every pipeline depth with one stream per batch)")
    ap.add_argument("--wide", default="auto", help="A/B: off = one pn2_linear per coarse-level layer; N = pn2_*_mlp_wide from N rows on")
    ap.add_argument("--no-hoist", action="store_true", help="A/B: FP first layers computed in place (pn2_fp_mlp_fused) instead of hoisted")
    ap.add_argument("--no-binned-bq", action="store_true", help="A/B: the sampler half of a pipelined batch does NOT bin the cloud for SA1's ball query")
    ap.add_argument("--no-coarse-geometry", action="store_true",
                    help="A/B: levels 2-4 through the separate sampling / ball-query / three_nn launches instead of pn2_coarse_geometry")
    ap.add_argument("--train", action="store_true",
                    help="BASELINE configs[3] instead of the headline: data-parallel TRAINING steps (forward with batch-stat "
                         "BN + weighted CE + backward + one flat RCCL gradient all-reduce + Adam), 16 scenes per GPU")
    ap.add_argument("--only-north-star", action="store_true",
                    help="run only the two north-star kernel measurements and print them (the command the rocprofv3 / PMC passes "
                         "of tools/gpu_round4.sh profile: profiles/r04_pmc_north_star.json)")
    ap.add_argument("--no-other-inputs", action="store_true", help="skip the S-randn / S-dup25 legs of the line")
    ap.add_argument("--stagger", default=None, metavar="B0,B1,..|off",
                    help="throughput regime on runtime.StaggeredPipeline: one stream per entry, the stream keeps that many sampled "
                         "batches ahead of its dense work (two graphs per batch on ONE stream; breaks the lockstep of a region that "
                         "starts with empty queues); off = one graph per batch on --pipeline streams (the r01-r04 execution)")
    args = ap.parse_args()
    if args.stagger is None:  # the default throughput execution, unless a script asks for the old one by naming --pipeline
        args.stagger = "off" if args.pipeline is not None else "0,0,1,1"
    if args.pipeline is None:
        args.pipeline = 4

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        launch_ranks(args)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%d: the line would not describe the job that ran" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py rank %d: LOCAL_RANK=%d but only %d GPU(s) visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_on = world > 1
    import pn2_amd as pn2
    if dist_on:
        import torch.distributed as dist
        pn2.dist.init_from_env(backend="nccl", device=dev)  # RCCL; same helper the gloo CPU test drives

    tfu = pn2.util.tf_util
    hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
    hp["batch_size"], hp["num_point"] = args.batch, args.points
    B, N = args.batch, args.points
    for kv in args.debug_set:
        what, value = kv.split("=")
        if not hasattr(pn2._lib._raw, "pn2_debug_set"):
            raise SystemExit("--debug-set needs a tuning build of the library: python open3d-pointnet2-semantic3d_amd/build.py --tuning")
        assert pn2._lib._raw.pn2_debug_set(int(what), int(value)) == 0
    if args.only_north_star:
        if rank == 0:
            print(json.dumps({"north_star": north_star_kernels(pn2, dev)}))
        return
    if args.train:
        bench_train(pn2, args, hp, B, N, rank, world, dev)
        if dist_on:
            dist.barrier()
            dist.destroy_process_group()
        return

    pc_np = s_scene(1000 + rank, B, N)        # each rank its own scenes (weak scaling)
    pc = torch.from_numpy(pc_np).to(dev)      # resident in HBM before the timed region
    store = tfu.set_default_store(tfu.VariableStore(device=dev, seed=0))  # replicated weights
    with torch.no_grad():
        pn2.model.get_sa_fp_features(pc, False, hp)  # creates variables
        g = torch.Generator().manual_seed(1)
        for k, v in store.params.items():             # non-trivial BN parameters
            if k.endswith("bn/gamma"):
                v.copy_((torch.rand(v.shape, generator=g) + 0.5).to(dev))
            elif k.endswith("bn/beta"):
                v.copy_((torch.randn(v.shape, generator=g) * 0.1).to(dev))
        for k, v in store.buffers.items():
            if k.endswith("moving_variance"):
                v.copy_((torch.rand(v.shape, generator=g) + 0.5).to(dev))

    pn2.util.pointnet_util.USE_HOISTED_FP = not args.no_hoist
    pn2.util.pointnet_util.USE_HOISTED_SA = not args.no_hoist

    if args.no_coarse_geometry:
        pn2.util.pointnet_util.USE_COARSE_GEOMETRY = False
    pn2._lib.lib.dup = tuple(args.dup)
    fused_fp = args.fp_front != "unfused"
    pn2.util.pointnet_util.USE_FUSED_FP = fused_fp
    if args.wide != "auto":  # A/B: the one-launch-per-level coarse MLPs (pn2_*_mlp_wide) on / off / from 4096 rows on
        pn2.util.pointnet_util.USE_MLP_WIDE = args.wide != "off"
        if args.wide.isdigit():
            pn2.util.pointnet_util.WIDE_MIN_ROWS = int(args.wide)

    def eager_on(x):
        with torch.no_grad():
            out, _ = pn2.model.get_sa_fp_features(x, False, hp)
        return out

    step = eager_step = lambda: eager_on(pc)  # noqa: E731
    pipe = None
    flush = lambda: None  # noqa: E731
    thr_inputs, n_streams, n_flight = [], 1, 1
    if not args.eager:
        # one hipGraph per forward: replay removes the ~40 Python-side launches from the step.
        # --pipeline P: P independent batches in flight (P graphs with their own buffers, replayed
        # round-robin on P streams), so one batch's latency-bound FPS (16 CUs) overlaps the MFMA
        # layers of the previous batch.  Every step is still one full forward over one batch.
        fwd = lambda x: pn2.model.get_sa_fp_features(x, False, hp)[0]  # noqa: E731
        mk = lambda n: (pc if n == 0 else torch.from_numpy(s_scene(2000 + 10 * rank + n, B, N)).to(dev))  # noqa: E731
        if args.stagger != "off":
            backlog = [int(v) for v in args.stagger.split(",")]
            pipe = pn2.runtime.StaggeredPipeline(lambda x: pn2.model.sa1_samples(x, hp, bins=not args.no_binned_bq),
                                                 lambda x, s: pn2.model.get_sa_fp_features(x, False, hp, sa1=s)[0], mk, backlog)
            caps = [pn2.runtime.CapturedForward(fwd, pc)]  # the latency regime: one graph, one batch in flight
            step = lambda: (lambda r: None if r is None else r[1])(pipe.step())  # noqa: E731

            def eager_on(x):  # the instrumented passes launch what the pipeline's two graphs hold  # noqa: F811
                with torch.no_grad():
                    return pn2.model.get_sa_fp_features(x, False, hp, sa1=pn2.model.sa1_samples(x, hp, bins=not args.no_binned_bq))[0]
            eager_step = lambda: eager_on(pc)  # noqa: E731
            thr_inputs, n_streams, n_flight = pipe.inputs(), pipe.P, pipe.batches_in_flight
        else:
            P = max(1, args.pipeline)
            caps = [pn2.runtime.CapturedForward(fwd, mk(n)) for n in range(P)]
            streams = [torch.cuda.Stream() for _ in range(P)]
            counter = [0]

            def step():
                i = counter[0] % P
                counter[0] += 1
                with torch.cuda.stream(streams[i]):
                    return caps[i].replay()
            thr_inputs, n_streams, n_flight = [c_.static_inputs[0] for c_ in caps], P, P
    for _ in range(args.warmup):
        step()
    flush()

    def timed_region():
        """EXACTLY args.steps steps, submitted and completed between two barrier + synchronize brackets -> (seconds, last output)"""
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = None
        for _ in range(args.steps):
            r_ = step()
            out = out if r_ is None else r_
        r_ = flush()       # (a staggered pipeline holds dense halves back: all of them are submitted before the synchronize)
        out = out if r_ is None else r_
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()
        return time.perf_counter() - t0, out

    # a K-step region is ~8 ms and starts with empty queues; ONE draw of it moved by +-3 % with the hardware-queue
    # mapping and the submission phase.  The headline is the MEDIAN of `--regions` such regions (each exactly K steps, each
    # bracketed as the contract says, each the maximum over ranks), after one untimed region; min / max / all of them are printed.
    timed_region()
    regions = []
    for _ in range(max(1, args.regions)):
        el, out = timed_region()
        per_rank = pn2.dist.gather_over_ranks(el, device=dev)
        regions.append((max(per_rank), per_rank))  # the slowest rank defines a region's time
    assert torch.isfinite(out).all()
    order = sorted(range(len(regions)), key=lambda i: regions[i][0])
    elapsed, per_rank_el = regions[order[(len(order) - 1) // 2]]  # the median region (lower median for an even count)
    per_rank_pps = [B * N * args.steps / v for v in per_rank_el]
    region_ms = [round(r_[0] / args.steps * 1e3, 4) for r_ in regions]
    streams_verified = getattr(pipe, "streams_verified_concurrent", None) if pipe is not None else None
    verified_per_rank = [None if v < 0 else int(v) for v in
                         pn2.dist.gather_over_ranks(-1 if streams_verified is None else streams_verified, device=dev)]

    # ---- latency regime: ONE batch in flight (what the reference's benchmark.py times), same K steps, bracketed the
    #      same way: replay -> synchronize per step
    latency_ms = None
    if not args.eager:
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            caps[0].replay()
            torch.cuda.synchronize()
        lat = time.perf_counter() - t1
        if dist_on:
            dist.barrier()
        latency_ms = pn2.dist.max_over_ranks(lat, device=dev) / args.steps * 1e3

    # ---- the same throughput regime over a 10x longer region (K = 20 steps is a 9 ms region that starts
    #      with empty queues; this figure says how far `value` is from the steady state) -- never `value`
    steady_ms = None
    if not args.eager and world == 1:
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(10 * args.steps):
            step()
        flush()
        torch.cuda.synchronize()
        steady_ms = (time.perf_counter() - t2) / (10 * args.steps) * 1e3

    # ---- instrumented pass: same steps, every launch bracketed by HIP events on its stream ----
    trace_steps = min(args.steps, 10)
    pn2._lib.lib.trace = []
    for _ in range(trace_steps):
        eager_step()  # eager launches (a graph replay cannot be bracketed per kernel); same kernels, same stream
    torch.cuda.synchronize()
    trace, pn2._lib.lib.trace = pn2._lib.lib.trace, None
    kernels = summarize_trace(trace, trace_steps)

    if rank == 0:
        total_points = world * B * N * args.steps
        ms_per_step = elapsed / args.steps * 1e3
        P_eff = 1 if args.eager else n_flight
        is_fps = lambda k: k["kernel"] in LATENCY_KERNELS  # noqa: E731
        dense = [k for k in kernels if not is_fps(k)]
        fps = [k for k in kernels if is_fps(k)]
        sum_all = sum(k["ms_per_step"] for k in kernels)
        sum_dense = sum(k["ms_per_step"] for k in dense)
        mfma = [k for k in kernels if k["bound"] == "mfma"]
        flops_step = sum(k["algorithmic_units"] * k["launches_per_step"] for k in mfma)
        # the timed regime decides which kernel set bounds the step: with >1 batch in flight the FPS chains (16 CUs
        # each) hide behind the other batches, the step is the serial sum of the chip-filling kernels
        timed = dense if P_eff > 1 else kernels
        dom = max(timed, key=lambda k: k["ms_per_step"])
        agg_tf = flops_step / (ms_per_step * 1e-3) / 1e12
        ref_flops = reference_flops(pn2, hp, B, N)
        eff_tf = ref_flops / (ms_per_step * 1e-3) / 1e12
        res = {
            "metric": "points/sec through SA+FP stack (B=16,N=8192)",
            "value": round(total_points / elapsed, 1), "unit": "points/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "value_statistic": "median of %d timed regions of exactly %d steps each (barrier + synchronize on both sides of every "
                               "region, maximum over ranks per region), after %d warm-up steps and one untimed region"
                               % (len(regions), args.steps, args.warmup),
            "timed_regions": len(regions), "ms_per_step_regions": region_ms,
            "value_min": round(total_points / max(r_[0] for r_ in regions), 1),
            "value_max": round(total_points / min(r_[0] for r_ in regions), 1),
            "streams_verified_concurrent": verified_per_rank,
            "value_regime": "throughput: %d independent B=%d batches in flight per GPU (see regimes.latency for one batch "
                            "in flight)" % (P_eff, B) if P_eff > 1 else "latency: one batch in flight",
            "config": {"workload": "configs[1]: full SSG PointNet++ SA x4 + FP x4 (semantic.json), inference forward, "
                                   "B=%d scenes x N=%d points xyz+rgb per GPU, fp32, S-scene synthetic input, "
                                   "random-init weights" % (B, N),
                       "batch_per_gpu": B, "num_point": N, "parallelism": "batch-sharded replicas x%d, no collective" % world,
                       "arith_mode": {"fps": int(pn2.config.fps_mode()), "ball_query": int(pn2.config.bq_mode()),
                                      "pinned_to": "oracle/_ref fast_noslp build of the reference's own kernels (contraction on)"},
                       "launch": "eager python launches" if args.eager else "one hipGraph replay per step",
                       "streams_per_batch": 1,
                       "execution": ("runtime.StaggeredPipeline: %d streams, two graphs per batch (SA1 sampling | the rest) on the "
                                     "batch's one stream, sampled batches kept ahead per stream %s" % (n_streams, args.stagger)
                                     if (args.stagger != "off" and not args.eager) else
                                     "one graph per batch, one stream per batch in flight"),
                       "streams": n_streams,
                       "batches_in_flight": P_eff,
                       "fp_front": "fused" if fused_fp else "materialised"},
            "regimes": {
                "throughput": {"batches_in_flight": P_eff, "ms_per_step": round(ms_per_step, 4),
                               "points_per_s": round(total_points / elapsed, 1)},
                "latency": None if latency_ms is None else {
                    "batches_in_flight": 1, "ms_per_step": round(latency_ms, 4),
                    "points_per_s": round(world * B * N / (latency_ms * 1e-3), 1),
                    "note": "replay + synchronize per step: what the reference's benchmark.py times"},
                "throughput_steady_state": None if steady_ms is None else {
                    "steps": 10 * args.steps, "ms_per_step": round(steady_ms, 4),
                    "points_per_s": round(B * N / (steady_ms * 1e-3), 1),
                    "note": "the throughput regime over a 10x longer timed region (not `value`)"}},
            "roofline": {"regime": "throughput" if P_eff > 1 else "latency",
                         "kernel": dom["kernel"], "args": dom["args"], "bound": dom["bound"],
                         "achieved": dom["achieved"], "peak": dom["peak"], "unit": dom["unit"], "frac": dom["frac"],
                         "traffic": pmc_traffic(dom["kernel"]), "traffic_source": PMC_FILE,
                         "algorithmic_units": dom["algorithmic_units"], "avg_us": dom["avg_us"],
                         "ms_per_step": dom["ms_per_step"],
                         "share_of_timed_kernels": round(dom["ms_per_step"] / max(1e-9, sum(k["ms_per_step"] for k in timed)), 3),
                         "flops_counted": "EXECUTED MFMA flops of the launch (a *_pre kernel runs its first layer only on the "
                                          "channels that were not hoisted onto the source rows; the hoisted product is its own "
                                          "linear launch in 'kernels')",
                         "aggregate": {"what": "all EXECUTED MFMA flops of one step / ms_per_step of the timed regime",
                                       "flops_per_step": int(flops_step), "achieved": round(agg_tf, 2),
                                       "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": round(agg_tf / MFMA_F32_PEAK_TF, 4)},
                         "aggregate_reference_formulation": {
                             "what": "flops the reference's formulation spends on the same step (SURVEY 8d: every conv on the "
                                     "grouped tensor) / ms_per_step: the rate a kernel set without hoisting would need",
                             "flops_per_step": int(ref_flops), "effective": round(eff_tf, 2), "unit": "TFLOP/s",
                             "frac_of_peak_equivalent": round(eff_tf / MFMA_F32_PEAK_TF, 4)},
                         "kernel_time_sums_ms": {"all": round(sum_all, 4), "without_fps": round(sum_dense, 4),
                                                 "fps": round(sum(k["ms_per_step"] for k in fps if k["kernel"] in FPS_KERNELS), 4),
                                                 "coarse_geometry": round(sum(k["ms_per_step"] for k in fps
                                                                              if k["kernel"] == "coarse_geometry"), 4),
                                                 "note": "without_fps = the chip-filling kernels: everything but the samplers "
                                                         "and pn2_coarse_geometry (latency-bound, a few workgroups per cloud)"}},
            "per_rank_points_per_s": [round(v, 1) for v in per_rank_pps],
            "kernels": kernels,
            "single_batch_latency_ms": None if latency_ms is None else round(latency_ms, 4),
            "gpu_ms_per_step_sum_of_kernels": round(sum_all, 4),
        }
        res.update(rccl_info(world))
        if fps:
            f0 = max([k for k in fps if k["kernel"] in FPS_KERNELS], key=lambda k: k["ms_per_step"])
            fb, fn, fm = f0["args"][:3]
            res["latency_limiter"] = {
                "kernel": f0["kernel"], "args": f0["args"], "avg_us": f0["avg_us"],
                "bound": "latency: m-1 dependent rounds on one CU per scene (neither HBM nor MFMA)",
                "ns_per_round": round(f0["avg_us"] * 1e3 / max(1, fm - 1), 1),
                "distance_evals_per_s": round(fb * (fm - 1) * fn / (f0["avg_us"] * 1e-6), 1),
                "fps_chain_ms": round(sum(k["ms_per_step"] for k in fps if k["kernel"] in FPS_KERNELS), 4),
                "geometry_chain_ms": round(sum_all - sum_dense, 4),  # samplers + pn2_coarse_geometry (levels 2-4 in one launch)
                "share_of_single_batch_latency": None if latency_ms is None else round((sum_all - sum_dense) / latency_ms, 3),
                "hbm_frac": f0["frac"],
                "note": "a round is a dependent chain (LDS read -> distance -> max tree -> 6 DPP steps -> LDS atomic -> "
                        "barrier -> read) of ~250-330 ns whatever the block shape (profiles/r02_fps_experiments.txt)"}
        if world == 1 and not args.eager and not args.no_other_inputs:
            # the data-dependent kernels (lazy FPS pruning, the grid ball query's >64-hit path, three_nn's
            # overflow rescan) on the reference benchmark's own input and on a duplicate-heavy cloud: SAME graphs (the
            # static input buffers are overwritten), same K, both regimes, per-kernel times from an eager pass
            res["other_inputs"] = {}
            base = {(k["kernel"], tuple(k["args"])): k["avg_us"] for k in kernels}
            for nm, gen in (("S-randn", s_randn), ("S-dup25", s_dup25)):
                arrs = [torch.from_numpy(gen(3000 + 10 * rank + i, B, N)).to(dev) for i in range(P_eff)]
                caps[0].static_inputs[0].copy_(arrs[0])
                for d_, a_ in zip(thr_inputs, arrs):
                    d_.copy_(a_)
                for _ in range(args.warmup):
                    step()
                flush()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                flush()
                torch.cuda.synchronize()
                thr = (time.perf_counter() - t0) / args.steps * 1e3
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    caps[0].replay()
                    torch.cuda.synchronize()
                lat = (time.perf_counter() - t0) / args.steps * 1e3
                pn2._lib.lib.trace = []
                for _ in range(trace_steps):
                    eager_on(arrs[0])
                torch.cuda.synchronize()
                tr, pn2._lib.lib.trace = pn2._lib.lib.trace, None
                ks = summarize_trace(tr, trace_steps)
                pick = lambda names: [{"args": k["args"], "avg_us": k["avg_us"]} for k in ks if k["kernel"] in names]  # noqa: E731
                slow = [{"kernel": k["kernel"], "args": k["args"], "avg_us": k["avg_us"],
                         "S-scene_avg_us": base[(k["kernel"], tuple(k["args"]))],
                         "ratio": round(k["avg_us"] / base[(k["kernel"], tuple(k["args"]))], 2)}
                        for k in ks if (k["kernel"], tuple(k["args"])) in base
                        and k["avg_us"] > 1.5 * base[(k["kernel"], tuple(k["args"]))] and k["avg_us"] > 5.0]
                res["other_inputs"][nm] = {
                    "ms_per_step": round(thr, 4), "points_per_s": round(B * N / (thr * 1e-3), 1),
                    "single_batch_latency_ms": round(lat, 4),
                    "fps_us": pick(FPS_KERNELS), "coarse_geometry_us": pick(("coarse_geometry",)), "query_ball_point_us": pick(("query_ball_point", "query_ball_point_binned")),
                    "three_nn_us": pick(("three_nn",)),
                    "gpu_ms_per_step_sum_of_kernels": round(sum(k["ms_per_step"] for k in ks), 4),
                    "slower_than_1.5x_S-scene": slow}
            caps[0].static_inputs[0].copy_(pc)  # back to S-scene for whatever follows
            for n_, d_ in enumerate(thr_inputs):
                d_.copy_(mk(n_))
        if not args.no_north_star:
            try:
                res["north_star"] = north_star_kernels(pn2, dev)
            except Exception as ex:  # keep the headline line alive
                res["north_star"] = {"error": repr(ex)}
        if world == 1 and not args.no_other_configs:
            res["other_configs"] = other_configs(pn2, dev, hp, args.steps)
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(pn2, store, pc_np, hp)
            except Exception as ex:
                res["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(res))
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
