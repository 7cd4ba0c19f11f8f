"""r05 experiment, measured and dropped: the throughput regime with the sampler half of every batch on dedicated SAMPLER streams
that run ahead of the dense streams (events between the halves), optionally with the chip split by CU masks
(hipExtStreamCreateWithCUMask: samplers on 16 / 32 CUs, dense kernels on the rest with persistent grids of 240 / 224).
Bit-identical outputs (it was tested like runtime.StaggeredPipeline is), and SLOWER than one graph per batch:

    ms per step, K = 20 / steady state (200 steps); one MI355X, configs[1]                 (profiles/r05_scheduling_study.txt)
    one graph per batch, 4 streams                                   0.423 / 0.384
    sampler-ahead 8 slots, 2 sampler + 2 dense streams               0.445 / 0.447
    sampler-ahead 8 slots, 1 + 3                                     0.462 / 0.421
    sampler-ahead 8 slots, 1 + 4 (5 streams on 4 hardware queues)    0.591 / 0.564
    sampler-ahead 8 slots, 4 + 4, GPU_MAX_HW_QUEUES=8                0.556 / 0.519
    CU split 16 | 240, 1 + 3, persistent grids of 240                0.471 / 0.413
    CU split 32 | 224, 2 + 2, persistent grids of 224                0.469 / 0.439
    both halves of a batch as two graphs on the batch's ONE stream   0.422 / 0.384   (two graphs per batch cost nothing)

What it established: (i) CU masks work, for eager launches and for graph replays alike, and mask bit i selects a CU of XCD i % 8
(tools/cu_mask_probe.py); (ii) the machine has four useful hardware queues -- a fifth stream shares one and a sampler chain then
blocks a dense stream; more queues (GPU_MAX_HW_QUEUES) are slower; (iii) the dense halves alone reach 0.370 ms per batch on three
streams, so in the steady state of the shipped execution (0.384) the sampling costs 0.015 ms per step: there is nothing left for a
scheduler to win there -- what one graph per batch loses is the START of a region (all streams in their sampler chains at once),
which runtime.StaggeredPipeline fixes without any cross-stream dependency.

Kept for the record; not imported by the package.  To run it again: copy the two definitions below next to StaggeredPipeline."""
import torch


def masked_stream(device, cu_bits):
    """A HIP stream whose kernels run only on the compute units whose bit is set in `cu_bits` (an int; on MI300-class parts
    bit i lands on XCD i % 8, so a run of 8k low bits is k CUs of every XCD) -> torch.cuda.ExternalStream.
    hipExtStreamCreateWithCUMask through ctypes; the stream lives until the process ends."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    nwords = max(1, (int(cu_bits).bit_length() + 31) // 32)
    words = (ctypes.c_uint32 * nwords)(*[(int(cu_bits) >> (32 * i)) & 0xFFFFFFFF for i in range(nwords)])
    st = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(nwords), words)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
    return torch.cuda.ExternalStream(st.value, device=device)


class SamplerAheadPipeline:
    """Throughput execution of a stack whose every batch starts with a long dependent chain on a few CUs (farthest point
    sampling: 16 workgroups for ~0.34 ms at semantic.json's shapes) followed by chip-filling dense work.

    A batch is TWO captured graphs: `sampler_fn(x) -> s` (the chain) and `dense_fn(x, s) -> y` (everything else).  Sampler
    graphs are replayed on sampler streams, dense graphs on dense streams, with one event between the halves of a batch and
    one guarding the reuse of a slot's buffers.  The sampler streams therefore run AHEAD of the dense streams by up to
    `slots` batches: the chain of batch k+1.. sits beside the dense layers of batch k from the first step of a timed region
    on, instead of all in-flight batches entering their chains together (one graph per batch on P streams starts every
    region in lockstep: P chains side by side on an idle chip, then P dense halves contending -- 0.43 ms per step over the
    first 20 steps against 0.38 in the steady state, r04).  Every step is still one full batch: both graphs of step k are
    launched by step(k) and nothing of a later step is computed before it was submitted.

    cu_split = (n_sampler_cus, n_total_cus) additionally confines the sampler streams to the first n_sampler_cus compute-unit
    bits and the dense streams to the rest (hipExtStreamCreateWithCUMask); None = ordinary streams."""

    def __init__(self, sampler_fn, dense_fn, batches, sampler_streams=2, dense_streams=2, warmup=2, cu_split=None,
                 same_stream=False):
        dev = batches[0].device
        self.slots = len(batches)
        self.inputs = [b.clone() for b in batches]
        if cu_split is None:
            mk_s = mk_d = lambda: torch.cuda.Stream(device=dev)  # noqa: E731
        else:
            ns, ntot = cu_split
            mk_s = lambda: masked_stream(dev, (1 << ns) - 1)  # noqa: E731
            mk_d = lambda: masked_stream(dev, ((1 << ntot) - 1) ^ ((1 << ns) - 1))  # noqa: E731
        self.d_streams = [mk_d() for _ in range(max(1, dense_streams))]
        # same_stream (diagnosis): both halves of a batch on its dense stream -- the cost of two graphs per batch by itself
        self.s_streams = self.d_streams if same_stream else [mk_s() for _ in range(max(1, sampler_streams))]
        self.same_stream = bool(same_stream)
        self.s_graphs, self.d_graphs, self.samples, self.outputs = [], [], [], []
        side = torch.cuda.Stream(device=dev)
        for x in self.inputs:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(warmup):
                    dense_fn(x, sampler_fn(x))
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            gs = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gs, capture_error_mode="thread_local"), torch.no_grad():
                s = sampler_fn(x)
            gd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gd, capture_error_mode="thread_local"), torch.no_grad():
                y = dense_fn(x, s)
            self.s_graphs.append(gs)
            self.d_graphs.append(gd)
            self.samples.append(s)
            self.outputs.append(y)
        self.sampled = [torch.cuda.Event() for _ in range(self.slots)]
        self.consumed = [torch.cuda.Event() for _ in range(self.slots)]
        self.used = [False] * self.slots
        self.count = 0

    def step(self, x=None):
        """submit the next batch (slot = step number mod slots; x = its input, or None: the batch resident in the slot's static
        input) -> its static output tensor (valid once the dense stream has finished it: torch.cuda.synchronize() or
        self.consumed[slot].synchronize(); overwritten `slots` steps later)"""
        k = self.count % self.slots
        self.count += 1
        ss = self.s_streams[k % len(self.s_streams)]
        ds = self.d_streams[k % len(self.d_streams)]
        if self.used[k]:
            ss.wait_event(self.consumed[k])  # the dense half of the slot's previous batch still reads the input and the samples
        if x is not None:
            ss.wait_stream(torch.cuda.current_stream(x.device))  # x was produced on the caller's stream
        with torch.cuda.stream(ss):
            if x is not None:
                self.inputs[k].copy_(x, non_blocking=True)
            self.s_graphs[k].replay()
            self.sampled[k].record(ss)
        ds.wait_event(self.sampled[k])
        with torch.cuda.stream(ds):
            self.d_graphs[k].replay()
            self.consumed[k].record(ds)
        self.used[k] = True
        return self.outputs[k]


