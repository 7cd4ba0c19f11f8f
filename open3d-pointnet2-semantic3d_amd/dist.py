"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm).

The SA/FP path shards over the batch dimension with NO data-path collective (every reference kernel
indexes batch by block; scenes are independent): inference = independent replicas.  Data-parallel
training (BASELINE config[3]: 16 scenes per GPU) adds ONE all-reduce per step of the flattened fp32
gradient (967,945 parameters = 3.87 MB for the semantic.json model): latency-bound over xGMI, so a
single bucket, no per-layer hooks.  BatchNorm statistics stay rank-local (the reference is a
single-GPU B=16 model; no sync-BN).

Everything here also runs on CPU with the gloo backend (tests/test_dist_cpu.py, world_size 2).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, device=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).
    -> (rank, world_size).  A world of 1 needs no process group."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_range(n_items, rank, world):
    """Contiguous [lo, hi) slice of n_items owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(value, device=None):
    """max of a python float over all ranks (bench: the slowest rank defines the step time)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value, device=None):
    """the python float of every rank, in rank order (bench: per-rank step times on the line, so a slow rank is visible)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


class FlatGradAllReduce:
    """Averages gradients across ranks with ONE all-reduce of a persistent flat fp32 bucket.

    usage:  bucket = FlatGradAllReduce(params);  loss.backward();  bucket.allreduce_();  opt.step()
    Parameters without a gradient contribute zeros (keeps every rank's bucket layout identical)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def nbytes(self):
        return self.numel * 4

    @torch.no_grad()
    def allreduce_(self):
        """Pack every .grad into the flat bucket (ONE multi-tensor copy, not a launch per parameter), all-reduce, and
        leave each p.grad as a view of the bucket: the optimizer reads the averaged gradient in place, no copy back.
        (The next backward allocates fresh .grad tensors after zero_grad(set_to_none=True), or accumulates straight
        into the views after zero_grad(set_to_none=False).)"""
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        src, dst = [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
        if dst:
            torch._foreach_copy_(dst, src)
        if world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(world)
        for p, v in zip(self.params, self.views):
            p.grad = v
        return self.flat


class OverlappedGradAllReduce:
    """Gradient all-reduce in TWO flat buckets so that the first one travels while the backward pass is still running.

    Backward produces gradients in reverse layer order: head and feature-propagation layers first, the set-abstraction
    layers last.  `params[split:]` (created last = differentiated first) form the EARLY bucket: a post-accumulate hook on
    each of them counts down, and when the last one lands the bucket is packed (one multi-tensor copy) and its
    all-reduce is launched asynchronously -- it overlaps the SA layers' backward.  `params[:split]` form the LATE
    bucket, reduced when backward has finished.  Both buckets are slices of ONE flat buffer (`flat`), so the
    optimizer still sees a single contiguous gradient.  Sums only: dividing by the world size is left to the optimizer
    (pn2_adam_step's grad_scale), which saves a pass over the buffer.

        bucket = OverlappedGradAllReduce(params, split);  bucket.begin();  loss.backward();  flat = bucket.finish()
    """

    def __init__(self, params, split):
        self.params = [p for p in params]
        self.split = int(split)
        self.numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for i, p in enumerate(self.params):
            if i == self.split:
                self.split_off = off
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        if self.split >= len(self.params):
            self.split_off = off
        self.late_flat, self.early_flat = self.flat[:self.split_off], self.flat[self.split_off:]
        self._pending, self._work, self.early_launched_in_backward = 0, None, False
        # defer_collectives: backward only PACKS the buckets; the caller reduces the whole flat buffer afterwards with
        # reduce_deferred() -- for a backward pass that is captured into a hipGraph (the collective stays outside)
        self.defer_collectives = False
        # skip_collectives: run the step as a lone rank would (bench: the no-communication step time of the SAME job, the
        # denominator of its scaling efficiency); the replicas diverge, so only after the measured region
        self.skip_collectives = False
        self._known_zero = set()
        self._expected_early = None
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params[self.split:]]

    def world(self):
        if self.skip_collectives:
            return 1
        return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1

    def _comm_on(self):
        return dist.is_available() and dist.is_initialized() and not self.skip_collectives

    @torch.no_grad()
    def time_collectives(self, iters=10):
        """each bucket's all-reduce ALONE (nothing to overlap with), ms per call: what the wire costs, to hold against the
        exposed time of a real step.  Sums garbage into `flat`: call it after the measured region only."""
        out = {"early_bytes": int(self.early_flat.numel()) * 4, "late_bytes": int(self.late_flat.numel()) * 4,
               "allreduce_early_ms": 0.0, "allreduce_late_ms": 0.0}
        if not self._comm_on() or dist.get_world_size() == 1:
            return out
        import time
        cuda = self.flat.is_cuda
        for key, buf in (("allreduce_early_ms", self.early_flat), ("allreduce_late_ms", self.late_flat)):
            if buf.numel() == 0:
                continue
            buf.zero_()
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)  # warm-up (connection set-up of this size)
            if cuda:
                torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(iters):
                dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            if cuda:
                torch.cuda.synchronize()
            out[key] = (time.perf_counter() - t0) / iters * 1e3
        return out

    def begin(self):
        """call before backward (after the gradients were reset)"""
        # early-bucket parameters that actually receive a gradient (learned from the first step: a bias in front of batch
        # norm never does, and must not keep the countdown from reaching zero)
        self._pending = self._expected_early if self._expected_early is not None else len(self.params) - self.split
        self._work, self.early_launched_in_backward = None, False

    def _pack(self, lo, hi):
        src, dst = [], []
        for i, (p, v) in enumerate(zip(self.params[lo:hi], self.views[lo:hi])):
            if p.grad is None:
                # no gradient (e.g. a bias in front of batch norm): its slice stays zero -- it was zero-filled when first
                # seen and nothing but an all-reduce of zeros has touched it since: no fill per step
                if (lo + i) not in self._known_zero:
                    v.zero_()
                    self._known_zero.add(lo + i)
            elif p.grad.data_ptr() != v.data_ptr():
                self._known_zero.discard(lo + i)
                src.append(p.grad)
                dst.append(v)
        if dst:
            torch._foreach_copy_(dst, src)

    @torch.no_grad()
    def _on_grad(self, p):
        self._pending -= 1
        if self._pending == 0:  # every early-bucket gradient has landed: pack and send while backward goes on
            self._pack(self.split, len(self.params))
            if self.world() > 1 and not self.defer_collectives:
                self._work = dist.all_reduce(self.early_flat, op=dist.ReduceOp.SUM, async_op=True)
            self.early_launched_in_backward = True

    @torch.no_grad()
    def finish(self):
        """after backward: reduce the late bucket, wait for the early one, leave every p.grad as a view of `flat`."""
        if self._expected_early is None:
            self._expected_early = sum(1 for p in self.params[self.split:] if p.grad is not None)
        if not self.early_launched_in_backward:  # first step, or a parameter's gradient did not arrive
            self._pack(self.split, len(self.params))
            if self.world() > 1 and not self.defer_collectives:
                self._work = dist.all_reduce(self.early_flat, op=dist.ReduceOp.SUM, async_op=True)
        self._pack(0, self.split)
        if self.world() > 1 and not self.defer_collectives:
            if self.split_off > 0:
                dist.all_reduce(self.late_flat, op=dist.ReduceOp.SUM)
            if self._work is not None:
                self._work.wait()
        for p, v in zip(self.params, self.views):
            p.grad = v
        return self.flat


    # ---- a backward pass run in two pieces (train.Trainer: three captured segments) -----------------------------------
    @torch.no_grad()
    def pack_early(self):
        """after the first piece (loss -> head -> FP): the early bucket's gradients into their slice of `flat`"""
        self._pack(self.split, len(self.params))

    @torch.no_grad()
    def pack_late_and_bind(self):
        """after the second piece (-> SA): the late bucket into `flat`; every p.grad becomes a view of `flat`"""
        self._pack(0, self.split)
        for p, v in zip(self.params, self.views):
            p.grad = v
        return self.flat

    def reduce_early_async(self):
        """the early bucket's all-reduce (sum), asynchronous: it travels while the second piece of backward runs"""
        if self._comm_on():
            return dist.all_reduce(self.early_flat, op=dist.ReduceOp.SUM, async_op=True)
        return None

    def reduce_late_and_wait(self, work):
        if self._comm_on() and self.split_off > 0:
            dist.all_reduce(self.late_flat, op=dist.ReduceOp.SUM)
        if work is not None:
            work.wait()
        return self.flat

    @torch.no_grad()
    def reduce_deferred(self):
        """the collective of a defer_collectives step: ONE all-reduce (sum) of the whole flat gradient on the current stream"""
        if self._comm_on():
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        return self.flat


def broadcast_parameters(tensors, src=0):
    """Make every rank start from rank `src`'s weights (replicated parameters)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in tensors:
            dist.broadcast(t.data if hasattr(t, "data") else t, src=src)
