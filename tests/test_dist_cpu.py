"""CPU tests of the N>1 path: world_size-2 gloo process groups exercising the exact helpers bench.py
and the data-parallel trainer use (batch sharding, max-over-ranks timing, flat gradient all-reduce)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import pn2_amd as pn2
    d = pn2.dist
    r, w = d.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    out = {}
    # 1. batch sharding: 128 scenes over 2 ranks = 64 each, 7 over 2 = 4 + 3, union is exact
    out["shard128"] = d.shard_range(128, rank, world)
    out["shard7"] = d.shard_range(7, rank, world)
    # 2. slowest rank defines the time
    out["tmax"] = d.max_over_ranks(1.0 + rank)
    # 3. flat-bucket gradient averaging == manual average, single bucket covers every parameter
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    d.broadcast_parameters(list(lin.parameters()))
    x = torch.full((4, 5), float(rank + 1))
    lin(x).sum().backward()
    local = [p.grad.clone() for p in lin.parameters()]
    bucket = d.FlatGradAllReduce(lin.parameters())
    bucket.allreduce_()
    out["numel"] = bucket.numel
    out["local"] = [g.tolist() for g in local]
    out["avg"] = [p.grad.tolist() for p in lin.parameters()]
    out["w0"] = lin[0].weight.detach().flatten().tolist()
    d.barrier()
    torch.distributed.destroy_process_group()
    q.put((rank, out))


@pytest.mark.timeout(180)
def test_world2_gloo_helpers():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=150) for _ in range(2))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert res[0]["shard128"] == (0, 64) and res[1]["shard128"] == (64, 128)
    assert res[0]["shard7"] == (0, 4) and res[1]["shard7"] == (4, 7)
    assert res[0]["tmax"] == res[1]["tmax"] == 2.0
    assert res[0]["w0"] == res[1]["w0"]  # replicated parameters after broadcast
    assert res[0]["numel"] == 5 * 7 + 7 + 7 * 3 + 3
    for a, b_, g0, g1 in zip(res[0]["avg"], res[1]["avg"], res[0]["local"], res[1]["local"]):
        ta, tb = torch.tensor(a), torch.tensor(b_)
        assert torch.equal(ta, tb)  # both ranks hold the same averaged gradient
        assert torch.allclose(ta, (torch.tensor(g0) + torch.tensor(g1)) / 2, rtol=1e-6, atol=1e-7)


def test_shard_range_partitions_exactly():
    import pn2_amd as pn2
    for n in (0, 1, 7, 16, 128, 1000):
        for w in (1, 2, 3, 8):
            parts = [pn2.dist.shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def test_training_schedules_match_reference_constants():
    """train.py:80-119: lr 1e-3 * 0.7^floor(step*16/200000) clipped at 1e-5; bn decay 0.5 -> 0.99."""
    import pn2_amd as pn2
    t = pn2.train
    assert t.learning_rate(0, 16) == 1e-3
    assert abs(t.learning_rate(12500, 16) - 7e-4) < 1e-12      # 12500*16 = 200000 -> one decay
    assert t.learning_rate(10 ** 7, 16) == 1e-5
    assert t.bn_decay(0, 16) == 0.5
    assert abs(t.bn_decay(12500, 16) - 0.75) < 1e-12
    assert t.bn_decay(10 ** 7, 16) == 0.99


def test_flat_bucket_aliases_grads_and_survives_both_zero_grad_modes():
    """FlatGradAllReduce (single process): after allreduce_ every p.grad is a view of the bucket (no copy back); a
    backward after zero_grad(set_to_none=True) packs fresh gradients, one after set_to_none=False accumulates straight
    into the views; a parameter without a gradient contributes zeros."""
    import pn2_amd as pn2
    torch.manual_seed(0)
    w1, w2, unused = (torch.nn.Parameter(torch.randn(*s)) for s in ((4, 3), (3,), (2, 2)))
    params = [w1, w2, unused]
    bucket = pn2.dist.FlatGradAllReduce(params)
    opt = torch.optim.SGD(params, lr=0.1)
    x = torch.randn(5, 4)
    for set_to_none in (True, False, True):
        opt.zero_grad(set_to_none=set_to_none)
        ((x @ w1 + w2) ** 2).sum().backward()
        ref1, ref2 = w1.grad.clone(), w2.grad.clone()
        flat = bucket.allreduce_()
        assert flat.numel() == 12 + 3 + 4
        for p, v in zip(params, bucket.views):
            assert p.grad.data_ptr() == v.data_ptr()
        assert torch.equal(w1.grad, ref1) and torch.equal(w2.grad, ref2) and not unused.grad.any()
        assert torch.equal(flat[:12].view(4, 3), ref1) and torch.equal(flat[12:15], ref2)
        before = w1.detach().clone()
        opt.step()
        assert torch.allclose(w1.detach(), before - 0.1 * ref1)


def _overlap_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import pn2_amd as pn2
    d = pn2.dist
    d.init_from_env(backend="gloo")
    torch.manual_seed(0)
    # "layer*" parameters first (late bucket: their gradients arrive last), then the head (early bucket)
    net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
    params = list(net.parameters())
    d.broadcast_parameters(params)
    bucket = d.OverlappedGradAllReduce(params, split=4)  # Linear0 + Linear1 late, Linear2 (head) early
    order = []
    for i, p in enumerate(params):
        p.register_post_accumulate_grad_hook(lambda _p, i=i: order.append((i, bucket.early_launched_in_backward)))
    out = {"steps": []}
    for step in range(2):
        for p in params:
            p.grad = None
        x = torch.full((5, 6), float(rank + 1 + step))
        bucket.begin()
        net(x).pow(2).sum().backward()
        local = [p.grad.clone() for p in params]
        flat = bucket.finish().clone()
        out["steps"].append({"local": [g.flatten().tolist() for g in local], "flat": flat.tolist(),
                             "early_in_backward": bucket.early_launched_in_backward})
        for p, v in zip(params, bucket.views):
            assert p.grad.data_ptr() == v.data_ptr()
    out["order"] = order
    out["split_off"] = bucket.split_off
    d.barrier()
    torch.distributed.destroy_process_group()
    q.put((rank, out))


@pytest.mark.timeout(180)
def test_world2_gloo_overlapped_two_bucket_allreduce():
    """The trainer's gradient exchange (dist.OverlappedGradAllReduce) on a world-2 gloo group: the early bucket (head) is
    packed and sent from inside backward, BEFORE the late layers' gradients exist; both ranks end with the same SUM of
    the local gradients in one flat buffer; every p.grad aliases it."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=150) for _ in range(2))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert res[0]["split_off"] == 6 * 8 + 8 + 8 * 8 + 8
    for step in range(2):
        a, b_ = res[0]["steps"][step], res[1]["steps"][step]
        assert a["early_in_backward"] and b_["early_in_backward"]
        assert a["flat"] == b_["flat"]  # identical on both ranks
        want = torch.cat([torch.tensor(x) + torch.tensor(y) for x, y in zip(a["local"], b_["local"])])
        assert torch.allclose(torch.tensor(a["flat"]), want, rtol=1e-6, atol=1e-6)  # SUM (the optimizer divides by the world size)
    # ordering: the head's gradients (params 4, 5) land first and trigger the early launch; when the late layers'
    # gradients (params 0..3) land, the early bucket is already on its way
    order = res[0]["order"][:6]
    assert sorted(i for i, _ in order[:2]) == [4, 5]
    assert all(flag for i, flag in order if i < 4)


def _two_piece_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import pn2_amd as pn2
    d = pn2.dist
    d.init_from_env(backend="gloo")
    torch.manual_seed(0)
    body = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 8), torch.nn.ReLU())   # "SA": late bucket
    head = torch.nn.Linear(8, 3)                                                                                   # "FP + head": early
    params = list(body.parameters()) + list(head.parameters())
    d.broadcast_parameters(params)
    bucket = d.OverlappedGradAllReduce(params, split=4)
    x = torch.full((5, 6), float(rank + 1))
    # reference: one backward pass
    head(body(x)).pow(2).sum().backward()
    local = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    # the backward pass in two pieces around a detached copy of the body's output (train.Trainer's three captured segments)
    feat = body(x)
    cut = feat.detach().requires_grad_(True)
    loss = head(cut).pow(2).sum()
    early = list(head.parameters())
    grads = torch.autograd.grad(loss, early + [cut])
    for p, g in zip(early, grads[:2]):
        p.grad = g
    bucket.pack_early()
    work = bucket.reduce_early_async()          # travels while the second piece runs
    late_missing = all(p.grad is None for p in body.parameters())
    torch.autograd.backward([feat], [grads[2]])
    bucket.pack_late_and_bind()
    flat = bucket.reduce_late_and_wait(work).clone()
    aliased = all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(params, bucket.views))
    d.barrier()
    torch.distributed.destroy_process_group()
    q.put((rank, {"local": [g.flatten().tolist() for g in local], "flat": flat.tolist(), "late_missing": late_missing,
                  "aliased": aliased}))


@pytest.mark.timeout(180)
def test_world2_gloo_backward_in_two_pieces_with_the_early_bucket_in_flight():
    """The multi-rank captured step's exchange (train.Trainer): backward cut at a detached copy of the body's
    output; the head's bucket is packed and its all-reduce launched asynchronously BEFORE the body's gradients exist, the
    body's bucket follows; both ranks end with the SUM of the one-piece gradients in the flat buffer."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_piece_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=150) for _ in range(2))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    a, b_ = res[0], res[1]
    assert a["late_missing"] and b_["late_missing"] and a["aliased"] and b_["aliased"]
    assert a["flat"] == b_["flat"]
    want = torch.cat([torch.tensor(x) + torch.tensor(y) for x, y in zip(a["local"], b_["local"])])
    assert torch.allclose(torch.tensor(a["flat"]), want, rtol=1e-6, atol=1e-6)


# ---- bench.py --gpus N really launches N ranks ------------------------------------------------------
def _run_bench(argv, env_extra=None, timeout=240):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, env=env, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.timeout(300)
def test_bench_gpus2_launches_two_ranks_and_prints_one_line():
    """Bare `bench.py --gpus 2` becomes the launcher: two processes rendezvous (gloo here, --dry-run = no GPU work), the
    timed region is barrier-bracketed, the slowest rank defines the step, rank 0 prints exactly one line with n_gpus 2."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["backend"] == "gloo" and line["steps"] == 3
    assert line["ms_per_step"] >= 2.0  # rank 1 sleeps 2 ms per step, rank 0 only 1 ms: max over ranks
    # every rank's own time is on the line (rank 1 is the slow one), as points/s for the inference bench
    assert len(line["per_rank_ms_per_step"]) == 2 and line["per_rank_ms_per_step"][1] > line["per_rank_ms_per_step"][0] >= 1.0
    assert len(line["per_rank_points_per_s"]) == 2 and line["per_rank_points_per_s"][0] > line["per_rank_points_per_s"][1] > 0


@pytest.mark.timeout(300)
def test_bench_train_line_decomposes_the_multi_rank_step():
    """`bench.py --train --gpus N` must let a sub-linear result be diagnosed from the record alone: each bucket's all-reduce
    alone, the time a step is exposed to the collectives, the same job's step with the collectives skipped, their ratio,
    and every rank's own step time (gloo world 2, --dry-run: the two-bucket exchange of dist.py on CPU tensors)."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run", "--train"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    for k in ("allreduce_early_ms", "allreduce_late_ms", "exposed_comm_ms", "ms_per_step_no_comm", "scaling_efficiency",
              "per_rank_ms_per_step", "early_bytes", "late_bytes"):
        assert k in line, k
    assert line["allreduce_early_ms"] > 0 and line["allreduce_late_ms"] > 0 and line["exposed_comm_ms"] >= 0
    assert line["early_bytes"] + line["late_bytes"] == 967945 * 4
    assert 0 < line["scaling_efficiency"] <= 1.0 and len(line["per_rank_ms_per_step"]) == 2


@pytest.mark.timeout(120)
def test_bench_never_reports_fewer_ranks_than_requested():
    """--gpus 2 on a node without two GPUs, or under a WORLD_SIZE that disagrees with --gpus: non-zero exit, no line."""
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this node really has two GPUs")
    r = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert r.returncode != 0 and "{" not in r.stdout
    assert "visible GPU" in (r.stderr + r.stdout)
    r = _run_bench(["--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run"],
                   env_extra=dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())))
    assert r.returncode != 0 and "{" not in r.stdout
    assert "WORLD_SIZE" in (r.stderr + r.stdout)
