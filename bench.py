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

from benchlib.accounting import (FPS_KERNELS, HBM_PEAK_GBS, LATENCY_KERNELS, MFMA_F32_PEAK_TF, PMC_FILE, kernel_model, pmc_traffic,  # noqa: E402,F401
                                 reference_flops, summarize_trace, time_call)
from benchlib.configs import both_regimes, north_star_kernels, other_configs  # noqa: E402,F401
from benchlib.inputs import s_dup25, s_randn, s_scene  # noqa: E402,F401
from benchlib.launch import dry_run, launch_ranks, rccl_info  # noqa: E402,F401
from benchlib.training import bench_train, train_comm_diagnosis  # noqa: E402,F401


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
