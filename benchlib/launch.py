"""One process per GPU: self-launch under torch.distributed.run, what RCCL reports about the job, and `--dry-run` (launcher,
rendezvous, barriers, max-over-ranks and the JSON line WITHOUT the GPU workload: tests/test_dist_cpu.py)."""
import json
import os
import subprocess
import sys
import time

import torch

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
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")] + sys.argv[1:]
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
