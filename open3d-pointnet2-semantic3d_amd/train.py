"""One data-parallel training step of the SSG network (BASELINE config[3]).

Restates what train.py of the reference does per batch (train.py:80-119 schedules, :381-388 Adam +
minimize, model.py:152-161 loss) on the MI355X layer API: forward with batch-statistic BatchNorm
(HIP index/gather kernels + differentiable torch layers), weighted sparse softmax cross-entropy,
backward through the HIP gradient kernels (group_point_grad, gather_point_grad,
three_interpolate_grad, pn2_linear_dgrad / _wgrad), a two-bucket RCCL all-reduce overlapped with backward, Adam.
"""
import torch

from . import dist as pdist
from . import model
from ._lib import check, lib, ptr, stream_ptr
from .util import tf_util


def learning_rate(step, batch_size, base_lr=1e-3, decay_step=200000, decay_rate=0.7, floor=1e-5):
    """tf.train.exponential_decay(staircase=True) clipped at 1e-5 (train.py:80-98)."""
    return max(base_lr * decay_rate ** ((step * batch_size) // decay_step), floor)


def bn_decay(step, batch_size, init=0.5, decay_step=200000, decay_rate=0.5, clip=0.99):
    """bn_decay = min(clip, 1 - init * rate^floor(step*B/decay_step)) (train.py:101-119)."""
    return min(clip, 1.0 - init * decay_rate ** ((step * batch_size) // decay_step))


def adam_lr_t(lr, t, beta1=0.9, beta2=0.999):
    """tf.train.AdamOptimizer's per-step rate: lr * sqrt(1 - beta2^t) / (1 - beta1^t), t = 1, 2, ..."""
    return lr * (1.0 - beta2 ** t) ** 0.5 / (1.0 - beta1 ** t)


class Trainer:
    """One process per GPU.  Everything between the input batch and the updated weights runs on the HIP library:
    forward GEMMs (pn2_linear), batch-norm kernels, weighted CE (pn2_weighted_ce_*), dropout (pn2_dropout), data and
    weight gradients (pn2_linear_dgrad / _wgrad), index-op gradients, and ONE Adam launch over flat parameter / moment
    buffers (pn2_adam_step).  torch supplies memory, streams, the autograd tape and torch.distributed.

    Step-dependent scalars live in device memory (learning rate with Adam's bias correction, dropout step), so after
    `warmup_eager` ordinary steps the step is captured and replayed: on one GPU as ONE hipGraph; with several ranks as three
    (forward + head/FP backward | SA backward | Adam) with the two buckets' all-reduces between them, the first one
    asynchronous so that it travels while the second graph replays -- no collective inside a captured region.  Eager steps
    (the warm-up, capture=False) launch the first bucket's all-reduce from inside backward (dist.OverlappedGradAllReduce)."""

    BETA1, BETA2, EPS = 0.9, 0.999, 1e-8  # tf.train.AdamOptimizer defaults (train.py:381-384)

    def __init__(self, hyperparams, num_class, store=None, device="cuda", capture=True, warmup_eager=3,
                 split_capture=None, overlap_collective=True):
        self.hp = dict(hyperparams)
        # schedule / optimizer keys of the reference's semantic.json (train.py:80-119, 380-386); its defaults when absent
        opt = str(self.hp.get("optimizer", "adam")).lower()
        if opt != "adam":
            raise ValueError("optimizer %r: only 'adam' (the reference's semantic.json setting, train.py:385-386) is built; "
                             "the 'momentum' branch of train.py:380-383 is not" % (opt,))
        self.sched = dict(base_lr=float(self.hp.get("learning_rate", 1e-3)), decay_step=int(self.hp.get("decay_step", 200000)),
                          lr_decay_rate=float(self.hp.get("learning_rate_decay_rate", 0.7)),
                          bn_init=float(self.hp.get("bn_init_decay", 0.5)), bn_rate=float(self.hp.get("bn_decay_decay_rate", 0.5)),
                          bn_clip=float(self.hp.get("bn_decay_clip", 0.99)))
        self.num_class = num_class
        self.store = store or tf_util.set_default_store(tf_util.VariableStore(device=device, seed=0))
        self.step_count = 0
        self.bucket = None
        self.capture, self.warmup_eager = bool(capture), int(warmup_eager)
        self._graph, self._graph_decay, self._static, self._stream = None, None, None, None
        self._static_geo, self._geo, self._geo_tag, self._geo_event, self._geo_stream = None, None, None, None, None
        # split_capture: the step as TWO graphs -- forward + backward (gradients packed, no collective inside), then Adam --
        # with the gradient all-reduce launched between them: how a step is captured when there are several ranks (a
        # collective inside a captured backward pass is avoided).  None = when world > 1; True forces it (tests).
        self.split_capture = split_capture
        # overlap_collective (with a split capture): THREE graphs -- forward + the backward pass down to the FP / SA boundary |
        # the SA part of the backward pass | Adam -- so that the all-reduce of the head + FP gradients (launched asynchronously
        # after the first) travels while the second replays; the SA bucket's all-reduce follows it.  False: two graphs around
        # ONE all-reduce of the whole flat gradient.
        self.overlap_collective = bool(overlap_collective)
        self._graph_adam, self._graph_late = None, None
        # staging (single-graph capture): the NEXT batch -- inputs, geometry, Adam's lr_t, the dropout step -- is put into staging
        # buffers by the geometry stream while this step runs; a one-node copy graph moves it into the step's static buffers.  The
        # trainer's stream then carries graphs only: an eager launch between two replays costs ~0.1 ms of idle queue (measured:
        # the captured step replays back to back in 3.33 ms, with one eager copy in between 3.45), a graph after a graph nothing.
        self._copy_graph, self._staging, self._staged_tag, self._copied_event = None, None, None, None
        # comm_events (bench.py --train): a list that receives, per captured multi-rank step, three HIP events on the
        # trainer's stream: early bucket launched | SA backward graph done | both buckets reduced
        self.comm_events = None
        self._world0 = None

    # ---- set-up ------------------------------------------------------------------------------------------------
    def _lazy_init(self, pc):
        # variables are created by the first forward (TF-style get_variable semantics).  That forward must leave no
        # trace: no autograd graph, and the moving averages restored (the reference applies ONE update per step,
        # train.py:381-388; the HIP BN kernel updates them through raw pointers even under no_grad).
        tf_util.set_default_store(self.store)
        with torch.no_grad():
            before = {k: v.clone() for k, v in self.store.buffers.items()}  # a pre-loaded store keeps its statistics
            model.get_model(pc[:1], True, self.num_class, self.hp, bn_decay=self._bn_decay(0, pc.shape[0]))
            for k, v in self.store.buffers.items():
                if k in before:
                    v.copy_(before[k])
                else:
                    v.fill_(0.0 if k.endswith("moving_mean") else 1.0)  # tf_util.py:571-581 initial moving averages
        names = list(self.store.params.keys())
        params = self.store.parameters()
        pdist.broadcast_parameters(params)
        pdist.broadcast_parameters(list(self.store.buffers.values()))
        # flat parameter / moment buffers: every parameter becomes a view of `flat_p` (same Parameter objects)
        dev = params[0].device
        n = sum(p.numel() for p in params)
        self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in params:
                v = self.flat_p[off:off + p.numel()].view_as(p)
                v.copy_(p)
                p.data = v
                off += p.numel()
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        # backward reaches the head and the FP layers first: they are the early bucket (created after layer1..layer4)
        split = next((i for i, k in enumerate(names) if not k.startswith("layer")), len(names))
        self.bucket = pdist.OverlappedGradAllReduce(params, split)
        # the gradient kernels write every parameter's gradient straight into its slice of the bucket's flat buffer
        # (tf_util.VariableStore.grad_view): backward leaves nothing to pack
        off = 0
        for p in self.bucket.params:
            self.store.grad_map[p.data_ptr()] = (self.bucket.flat, off, p.numel())
            off += p.numel()
        # pn2_adam_step's device-side scalars: [lr_t, beta1, beta2, eps, grad_scale].  Only lr_t changes per step; it is
        # written with a fill whose value travels BY VALUE in the launch (no host buffer a run-ahead host could rewrite
        # before an asynchronous copy has read it: with sync=False the host is several steps ahead of the device).
        self.hyper = torch.tensor([0.0, self.BETA1, self.BETA2, self.EPS, 1.0 / self.bucket.world()], dtype=torch.float32).to(dev)
        self._lr_slot = self.hyper[0:1]
        self._world0 = self.bucket.world()  # frozen into grad_scale above: a process group created later would mis-scale
        self._stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._geo_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None

    def _learning_rate(self, step, batch_size):
        c = self.sched
        return learning_rate(step, batch_size, c["base_lr"], c["decay_step"], c["lr_decay_rate"])

    def _bn_decay(self, step, batch_size):
        c = self.sched
        return bn_decay(step, batch_size, c["bn_init"], c["decay_step"], c["bn_rate"], c["bn_clip"])

    # ---- one step ------------------------------------------------------------------------------------------------
    def _forward_backward(self, pc, labels, smpw, decay, geometry=None):
        if self.store.zero_arena is None:
            self.store.zero_arena = tf_util.ZeroArena(self.flat_p.device)  # first step: measures what the step needs
        elif self.store.zero_arena.buf is None:
            self.store.zero_arena.allocate()
        self.store.zero_arena.reset()  # ONE zero fill for every accumulator of the step
        self.bucket.flat.zero_()       # and one for the flat gradient the weight-gradient kernels add into
        for p in self.bucket.params:
            p.grad = None
        self.store.grad_direct = True
        try:
            logits, _ = model.get_model(pc, True, self.num_class, self.hp, bn_decay=decay, geometry=geometry)
            loss = model.get_loss(logits, labels, smpw)
            self.bucket.begin()
            loss.backward()
        finally:
            self.store.grad_direct = False
        return loss.detach(), self.bucket.finish()

    def _forward_backward_early(self, pc, labels, smpw, decay, geometry=None):
        """first piece of a cut step: forward, loss, and the backward pass of the head and the FP modules down to (detached
        copies of) the SA outputs; the early bucket is packed.  -> loss"""
        if self.store.zero_arena is None:
            self.store.zero_arena = tf_util.ZeroArena(self.flat_p.device)
        elif self.store.zero_arena.buf is None:
            self.store.zero_arena.allocate()
        self.store.zero_arena.reset()
        self.bucket.flat.zero_()
        for p in self.bucket.params:
            p.grad = None
        self.store.grad_direct = True
        try:
            logits, ep = model.get_model(pc, True, self.num_class, self.hp, bn_decay=decay, geometry=geometry, cut_sa_fp=True)
            loss = model.get_loss(logits, labels, smpw)
            early = list(self.bucket.params[self.bucket.split:])
            cut = list(ep["sa_features_cut"])
            grads = torch.autograd.grad(loss, early + cut, allow_unused=True)
        finally:
            self.store.grad_direct = False
        for p, g in zip(early, grads[:len(early)]):
            p.grad = g
        self.bucket.pack_early()
        self._cut = [(t, g) for t, g in zip(ep["sa_features"][1:], grads[len(early):]) if g is not None]
        return loss.detach()

    def _backward_late(self):
        """second piece: the SA modules' backward pass from the gradients at the cut; the late bucket is packed"""
        self.store.grad_direct = True
        try:
            torch.autograd.backward([t for t, _ in self._cut], [g for _, g in self._cut])
        finally:
            self.store.grad_direct = False
        self._cut = None
        return self.bucket.pack_late_and_bind()

    def _adam(self, flat_g):
        with torch.cuda.device(self.flat_p.device):
            check(lib.pn2_adam_step(self.flat_p.numel(), ptr(self.flat_p), ptr(flat_g), ptr(self.flat_m), ptr(self.flat_v),
                                    ptr(self.hyper), stream_ptr()), "pn2_adam_step")

    def _step_body(self, pc, labels, smpw, decay, geometry=None):
        loss, flat_g = self._forward_backward(pc, labels, smpw, decay, geometry)
        self._adam(flat_g)
        return loss

    # ---- geometry prefetch -----------------------------------------------------------------------------------------
    def _xyz_of(self, pc):
        return pc[:, :, 0:3].contiguous() if self.hp["use_color"] else pc.contiguous()

    @staticmethod
    def _tag(pc):
        return (pc.data_ptr(), pc._version, tuple(pc.shape))

    def _geometry_for(self, pc, caller):
        """geometry of THIS batch: taken from the prefetch when `pc` is the tensor announced as `next_pc` by the previous
        call, computed now otherwise.  Returns with `caller` (a stream) ordered after its completion."""
        tag = (pc.data_ptr(), pc._version, tuple(pc.shape))
        if self._geo is not None and self._geo_tag == tag:
            caller.wait_event(self._geo_event)
            geo = self._geo
            for t in model.geometry_tensors(geo):
                t.record_stream(caller)  # allocated on the side stream, consumed on this one
        else:
            geo = model.compute_geometry(self._xyz_of(pc), self.hp, plans=True)
        self._geo, self._geo_tag = None, None
        return geo

    def _prefetch(self, next_pc, after_event, next_labels=None, next_smpw=None):
        """launch the geometry of the NEXT batch on the side stream; it may start once `after_event` has passed (the
        consumer of the previous prefetch has taken its copy).  With the whole next batch announced (labels, weights) and a
        captured step with staging buffers, the batch, its geometry and the next step's scalars are also put into the staging
        buffers there: the next call then launches graphs only."""
        self._staged_tag = None
        if next_pc is None:
            return
        g = self._geo_stream
        g.wait_event(after_event)
        with torch.cuda.stream(g):
            self._geo = model.compute_geometry(self._xyz_of(next_pc), self.hp, plans=True)
            stg = self._staging
            if (stg is not None and next_labels is not None and next_smpw is not None and self._copied_event is not None
                    and all(a.shape == d.shape and a.dtype == d.dtype and a.is_contiguous()
                            for a, d in zip((next_pc, next_labels, next_smpw), stg["inputs"]))):
                g.wait_event(self._copied_event)  # the copy graph that read the staging buffers last has run
                srcs = [next_pc, next_labels, next_smpw] + model.geometry_tensors(self._geo)
                dsts = stg["inputs"] + stg["geo"]
                if len(srcs) == len(dsts):
                    nxt = self.step_count + 1  # the step this batch is for (0-based), its Adam time step is nxt + 1
                    lr_t = adam_lr_t(self._learning_rate(nxt, next_pc.shape[0]), nxt + 1, self.BETA1, self.BETA2)
                    fills = [(stg["lr"], lr_t)] + [(t, nxt) for t in stg["steps"]]
                    tf_util.multi_copy_(dsts, srcs, fills=fills)
                    self._staged_tag = (self._tag(next_pc), self._tag(next_labels), self._tag(next_smpw), nxt)
            self._geo_event = torch.cuda.Event()
            self._geo_event.record(g)
        self._geo_tag = (next_pc.data_ptr(), next_pc._version, tuple(next_pc.shape))

    def prefetch_geometry(self, next_pc):
        """Announce the next batch AFTER the current step was enqueued (e.g. when it is itself produced on another stream,
        examples/train_synthetic.py): its geometry chain starts once the work queued so far on the CURRENT stream is done."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._prefetch(next_pc, ev)

    def train_step(self, pc, labels, smpw, sync=True, next_pc=None, next_labels=None, next_smpw=None):
        """pc (B,N,6) float32, labels (B,N) int, smpw (B,N) float32 -> loss (python float; the device scalar when
        sync=False, to be read after the next synchronisation point).
        next_pc: the point cloud of the NEXT call (optional).  Its FPS / ball-query / three_nn chain -- weight-independent,
        latency-bound, 16 of 256 CUs -- then runs on a side stream beside this step's dense work instead of in front of
        the next one (what the reference's mp.Pool data loader does for its CPU pre-processing).  next_labels / next_smpw
        (optional, with next_pc): the rest of the next batch; a captured step then finds the whole batch staged and launches
        graphs only (see _prefetch)."""
        tf_util.set_default_store(self.store)
        if self.bucket is None:
            self._lazy_init(pc)
        b = pc.shape[0]
        world = self.bucket.world()
        if world != self._world0 and not self.bucket.skip_collectives:
            raise RuntimeError("the process group changed after the trainer's first step (world %d -> %d): the gradient scale "
                               "1/world is part of the optimizer state; build a new Trainer" % (self._world0, world))
        t = self.step_count + 1
        lr = self._learning_rate(self.step_count, b)
        decay = self._bn_decay(self.step_count, b)
        lr_t = adam_lr_t(lr, t, self.BETA1, self.BETA2)
        use_graph = self.capture and pc.is_cuda and self.step_count >= self.warmup_eager
        split = use_graph and (world > 1 if self.split_capture is None else bool(self.split_capture))
        caller = torch.cuda.current_stream()
        if not use_graph:
            self._lr_slot.fill_(lr_t)
            self.store.set_step(self.step_count)
            geo = self._geometry_for(pc, caller)
            taken = torch.cuda.Event()
            taken.record(caller)
            self._prefetch(next_pc, taken)  # (an eager step has no staging buffers)
            loss = self._step_body(pc, labels, smpw, decay, geometry=geo)
        else:
            # Replays and the per-step writes they depend on run on the trainer's OWN stream.  Launching the graph into
            # the null stream is not safe on this stack: work queued on the null stream after hipGraphLaunch started
            # before the graph had finished (the next step's input copy then faulted with "write access to a read-only
            # page" around the 13th replay; a device-wide synchronise in between hid it).
            self._stream.wait_stream(caller)
            with torch.cuda.stream(self._stream):
                recapture = self._graph is None or self._graph_decay != decay or self._static[0].shape != pc.shape
                staged = (not recapture and self._copy_graph is not None and self._staged_tag is not None and self._staged_tag ==
                          (self._tag(pc), self._tag(labels), self._tag(smpw), self.step_count))
                if staged:
                    # the geometry stream has put this batch, its geometry and this step's scalars into the staging buffers:
                    # copy graph -> step graph, no eager launch on this stream
                    self._stream.wait_event(self._geo_event)
                    self._geo, self._geo_tag, self._staged_tag = None, None, None
                    self._copy_graph.replay()
                    self._copied_event = torch.cuda.Event()
                    self._copied_event.record(self._stream)
                    self._prefetch(next_pc, self._copied_event, next_labels, next_smpw)
                    self._graph.replay()
                    loss = self._static[3]
                    if sync:
                        loss = float(loss)
                    caller.wait_stream(self._stream)
                    self.step_count += 1
                    self.store.train_epoch += 1
                    return float(loss) if sync else loss
                geo = self._geometry_for(pc, self._stream)  # eager (or eagerly prefetched: prefetch_geometry)
                if recapture:
                    self._capture(pc, labels, smpw, decay, geo, split)
                # every per-step device-to-device copy (inputs, and this batch's geometry when it is not in place yet) AND the
                # step's scalars (Adam's lr_t, the dropout step: their values travel in the launch arguments) in ONE launch: they
                # sit between two graphs on the critical path
                pairs = [(d, s_) for d, s_ in zip(self._static[:3], (pc, labels, smpw))
                         if d.data_ptr() != s_.data_ptr() and s_.is_contiguous() and s_.dtype == d.dtype]
                for d, s_ in zip(self._static[:3], (pc, labels, smpw)):
                    if d.data_ptr() != s_.data_ptr() and not (s_.is_contiguous() and s_.dtype == d.dtype):
                        d.copy_(s_, non_blocking=True)
                pairs += list(zip(model.geometry_tensors(self._static_geo), model.geometry_tensors(geo)))
                fills = [(self._lr_slot, lr_t)] + self.store.step_fills(self.step_count)
                if pairs and len(fills) <= 4:
                    tf_util.multi_copy_([d for d, _ in pairs], [s_ for _, s_ in pairs], fills=fills)
                else:
                    self._lr_slot.fill_(lr_t)
                    self.store.set_step(self.step_count)
                    if pairs:
                        tf_util.multi_copy_([d for d, _ in pairs], [s_ for _, s_ in pairs])
                taken = torch.cuda.Event()
                taken.record(self._stream)
                self._copied_event = taken  # (nothing reads the staging buffers in this form of the step)
                self._prefetch(next_pc, taken, next_labels, next_smpw)
                self._graph.replay()
                if self._graph_late is not None:  # three segments: the early bucket travels while the SA backward replays
                    ev = self.comm_events  # bench: [(early launched, SA backward done, both buckets reduced)] per step
                    if ev is not None:
                        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                        e0.record(self._stream)
                    work = self.bucket.reduce_early_async()
                    self._graph_late.replay()
                    if ev is not None:
                        e1.record(self._stream)
                    self.bucket.reduce_late_and_wait(work)
                    if ev is not None:
                        e2.record(self._stream)
                        ev.append((e0, e1, e2))
                    self._graph_adam.replay()
                elif self._graph_adam is not None:  # split capture: the collective runs between the two graphs
                    self.bucket.reduce_deferred()
                    self._graph_adam.replay()
                loss = self._static[3]
                if sync:
                    loss = float(loss)
            caller.wait_stream(self._stream)
        self.step_count += 1
        # A replayed graph updates parameters and moving averages through raw pointers: neither a tensor version counter
        # nor the training-mode layer calls (which only run while capturing) tell the inference-weight cache
        # (VariableStore.folded) that they changed.  Every step does.
        self.store.train_epoch += 1
        return float(loss) if sync else loss

    def _capture(self, pc, labels, smpw, decay, geo, split=False):
        """record forward + loss + backward + Adam of one step into a hipGraph (static input and geometry buffers); split:
        Adam in a graph of its own, the gradient all-reduce is launched between the two replays."""
        torch.cuda.synchronize()
        st = [pc.clone(), labels.clone(), smpw.clone()]
        sg = model.clone_geometry(geo, self._xyz_of(st[0]))
        for p in self.bucket.params:
            p.grad = None
        # capture_error_mode thread_local: with a process group alive, RCCL's watchdog thread polls the events of earlier
        # collectives (hipEventQuery); under the default GLOBAL mode that call is illegal while ANY thread captures and takes the
        # process down (seen as an abort in destroy_process_group after a full test session)
        mode = dict(capture_error_mode="thread_local")
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, **mode):
            if split and self.overlap_collective:
                loss = self._forward_backward_early(st[0], st[1], st[2], decay, geometry=sg)
            elif split:
                self.bucket.defer_collectives = True
                try:
                    loss, flat_g = self._forward_backward(st[0], st[1], st[2], decay, geometry=sg)
                finally:
                    self.bucket.defer_collectives = False
            else:
                loss = self._step_body(st[0], st[1], st[2], decay, geometry=sg)
        self._graph_adam, self._graph_late = None, None
        if split and self.overlap_collective:
            gl = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gl, pool=g.pool(), **mode):
                flat_g = self._backward_late()
            self._graph_late = gl
        if split:
            ga = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga, **mode):
                self._adam(flat_g)
            self._graph_adam = ga
        self._graph, self._graph_decay, self._static, self._static_geo = g, decay, st + [loss], sg
        # staging buffers + the one-node copy graph (single-graph capture only: a multi-rank step has collectives between its
        # graphs anyway)
        self._copy_graph, self._staging, self._staged_tag = None, None, None
        if not split:
            steps = [t[1:2] for t in self.store._dropout.values()]
            dsts = st + model.geometry_tensors(sg) + [self._lr_slot] + steps
            stg = {"inputs": [t.clone() for t in st], "geo": [t.clone() for t in model.geometry_tensors(sg)],
                   "lr": self._lr_slot.clone(), "steps": [t.clone() for t in steps]}
            srcs = stg["inputs"] + stg["geo"] + [stg["lr"]] + stg["steps"]
            gc = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gc, **mode):
                tf_util.multi_copy_(dsts, srcs)
            self._copy_graph, self._staging = gc, stg
        # the capture itself executed nothing: the replay that follows is this step
