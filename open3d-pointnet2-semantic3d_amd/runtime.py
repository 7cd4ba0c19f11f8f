"""HIP-graph capture of a forward pass.

A PointNet++ forward is ~40 short launches; issued from Python the host becomes the bottleneck
(measured: 2.7 ms wall for 2.0 ms of kernels).  `CapturedForward` records the whole pass -- the
ctypes launches of libpn2_hip.so go to torch's current stream, which is the capture stream -- into
one hipGraph and replays it per step.  Inputs are copied into static buffers; outputs are static
tensors overwritten by each replay.
"""
import torch


class CapturedForward:
    def __init__(self, fn, *example_inputs, warmup=3):
        """fn(*tensors) -> tensor | tuple of tensors; shapes/dtypes are frozen at capture time."""
        self.static_inputs = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):  # creates variables, folded weights, sets kernel attributes
                fn(*self.static_inputs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread-local error mode: with a process group alive (bench.py --gpus N) RCCL's watchdog thread may poll events while
        # this thread captures; in the default global mode that is an error for the whole process
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"), torch.no_grad():
            self.static_outputs = fn(*self.static_inputs)

    def __call__(self, *inputs):
        for dst, src in zip(self.static_inputs, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.static_outputs

    def replay(self):
        """replay on the inputs already resident in the static buffers"""
        self.graph.replay()
        return self.static_outputs


def concurrent_streams(n, device=None, candidates=12, spin_cycles=300000, strict=False):
    """-> (streams, verified): n torch streams and how many of them the hardware was SEEN to run side by side.
    HIP multiplexes streams onto a few hardware queues (4 by default) in an order that depends on what the process submitted
    before; two streams that land on one queue serialise, and a 4-stream pipeline then runs like a 2- or 3-stream one
    (measured: 0.505 instead of 0.383 ms per step for the FIRST four streams of a process, profiles/r05_scheduling_study.txt #8).
    So: draw `candidates` streams, time a spin kernel on a growing set of them, and keep a stream only if the set's wall time
    stays that of ONE spin (best of three runs < 1.5 x; a serialised pair costs 2 x, so the threshold sits midway and a noisy
    box errs towards rejecting a stream, never towards accepting a serialised one: the minimum of repeated runs can only be
    inflated by noise, not deflated).  When fewer than n concurrent streams are found the set is filled up with unverified
    candidates -- the pipeline still works, a hardware queue is shared -- and `verified` < n says so (strict=True: raise
    instead).  Candidates that are not returned are dropped (torch returns their HIP streams to its pool)."""
    import time
    cand = [torch.cuda.Stream(device=device) for _ in range(max(n, candidates))]

    def wall(streams):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for st in streams:
            with torch.cuda.stream(st):
                torch.cuda._sleep(spin_cycles)
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0

    wall(cand[:1])  # first use of the spin kernel (module load)
    t1 = min(wall(cand[:1]) for _ in range(5))
    chosen = [cand[0]]
    for st in cand[1:]:
        if len(chosen) == n:
            break
        # best of five (r06: best of three rejected a good stream once in three suite runs on a box still draining other work)
        if min(wall(chosen + [st]) for _ in range(5)) < 1.5 * t1:
            chosen.append(st)
    verified = len(chosen)
    if verified < n and strict:
        raise RuntimeError("concurrent_streams: only %d of %d streams run side by side on this device" % (verified, n))
    for st in cand:  # not enough concurrent ones: fill up (the pipeline still works, a queue is shared)
        if len(chosen) < n and st not in chosen:
            chosen.append(st)
    del cand
    return chosen, verified


class StaggeredPipeline:
    """Throughput execution of a stack whose every batch starts with a long dependent chain on a few CUs (farthest point
    sampling: 16 workgroups for ~0.34 ms at semantic.json's shapes) followed by chip-filling dense work.

    P batch streams; every batch is TWO captured graphs -- sampler_fn(x) -> s (the chain) and dense_fn(x, s) -> y (everything
    else) -- launched on the batch's ONE stream (no events, no cross-stream dependencies; two graphs per batch on one stream cost
    what one graph costs, measured), but not in the same phase on every stream: stream i keeps backlog[i] sampled batches ahead of
    its dense work,
        backlog 0:  F D F D F D ...                      backlog 1:  F F D F D F D ... D
    so that the streams of a region that starts with empty queues do not all sit in their sampler chains together and then all
    contend in their dense halves.  One graph per batch on 4 streams does exactly that for the first ~20 steps (0.423 ms per step
    at K = 20 against 0.384 in its steady state); backlogs (0, 0, 1, 1) give 0.408-0.412 at K = 20 and the same steady state
    (profiles/r05_scheduling_study.txt).  step() submits the sampler half of the next batch and the dense half of the batch
    `backlog` steps older on that stream; flush() submits the dense halves still held back -- call it before synchronising:
    every step's work is then submitted, and completed, inside the timed region, and nothing of a step is computed before the
    step was submitted."""

    def __init__(self, sampler_fn, dense_fn, make_batch, backlog=(0, 0, 1, 1), warmup=2, streams=None, strict_streams=False):
        """make_batch(n) -> the example input of slot n (sum(backlog) + len(backlog) slots, cloned into static buffers).
        streams: the caller's own streams (one per backlog entry; `streams_verified_concurrent` is then None: not probed);
        default: concurrent_streams(P), whose verified count is kept in `streams_verified_concurrent` (bench.py prints it per
        rank) -- strict_streams=True raises when fewer than P streams were seen to run side by side."""
        self.backlog = [int(b) for b in backlog]
        if not self.backlog or min(self.backlog) < 0:
            raise ValueError("backlog: one non-negative entry per stream")
        self.P = len(self.backlog)
        if streams is not None:
            if len(streams) != self.P:
                raise ValueError("streams: one per backlog entry")
            self.streams, self.streams_verified_concurrent = list(streams), None
        else:
            self.streams, self.streams_verified_concurrent = concurrent_streams(self.P, strict=strict_streams)
        self.slots = []   # per stream: [static input, sampler graph, dense graph, static output, samples]
        side = torch.cuda.Stream()
        n = 0
        for i in range(self.P):
            row = []
            for _ in range(self.backlog[i] + 1):
                x = make_batch(n).clone()
                n += 1
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side), torch.no_grad():
                    for _ in range(warmup):  # creates variables, folded weights, sets kernel attributes
                        dense_fn(x, sampler_fn(x))
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                gs = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gs, capture_error_mode="thread_local"), torch.no_grad():
                    s = sampler_fn(x)
                gd = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gd, capture_error_mode="thread_local"), torch.no_grad():
                    y = dense_fn(x, s)
                row.append((x, gs, gd, y, s))
            self.slots.append(row)
        self.next_slot = [0] * self.P
        self.pending = [[] for _ in range(self.P)]
        self.count = 0

    @property
    def batches_in_flight(self):
        return sum(self.backlog) + self.P

    def step(self, x=None):
        """submit the next batch (x = its input, or None: the batch resident in the slot's static buffer).  -> (slot, output)
        of the dense half this call submitted -- the batch `backlog` steps older on this stream -- or None when it only
        sampled.  An output is valid once its stream has run it (stream.synchronize / torch.cuda.synchronize) and is
        overwritten when its slot comes round again (backlog + 1 steps of that stream later)."""
        i = self.count % self.P
        self.count += 1
        k = self.next_slot[i]
        self.next_slot[i] = (k + 1) % len(self.slots[i])
        st = self.streams[i]
        if x is not None:
            st.wait_stream(torch.cuda.current_stream(x.device))  # x was produced on the caller's stream
        with torch.cuda.stream(st):
            if x is not None:  # (the slot's previous batch is done with the buffer: its dense half precedes us on this stream)
                self.slots[i][k][0].copy_(x, non_blocking=True)
            self.slots[i][k][1].replay()
            self.pending[i].append(k)
            if len(self.pending[i]) > self.backlog[i]:
                j = self.pending[i].pop(0)
                self.slots[i][j][2].replay()
                return (i, j), self.slots[i][j][3]
        return None

    def flush(self):
        """submit the dense halves that are still held back -> [((stream, slot), output)] in submission order"""
        done = []
        for i in range(self.P):
            with torch.cuda.stream(self.streams[i]):
                while self.pending[i]:
                    j = self.pending[i].pop(0)
                    self.slots[i][j][2].replay()
                    done.append(((i, j), self.slots[i][j][3]))
        return done

    def inputs(self):
        """the static input buffers, slot order (= make_batch order)"""
        return [sl[0] for row in self.slots for sl in row]
