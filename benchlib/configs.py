"""The kernel-level north-star targets of BASELINE.json at their own shape, and BASELINE configs[2] (MSG module) / configs[4]
(large-scene layer) / configs[3]@1gpu timed like the headline: one batch in flight and the staggered throughput pipeline."""
import time

import numpy as np
import torch

from .accounting import HBM_PEAK_GBS, MFMA_F32_PEAK_TF, time_call
from .inputs import s_scene

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
