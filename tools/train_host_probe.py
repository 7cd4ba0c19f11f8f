"""Where a training step's wall time goes on the host: per-call enqueue time of Trainer.train_step(sync=False) against the
step time, with and without the geometry prefetch, and the graph replay alone.  Run on the GPU box."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pn2_amd as pn2
from bench import s_scene

dev = torch.device("cuda:0")
B, N = 16, 8192
hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
rs = np.random.RandomState(100)
pc = torch.from_numpy(np.concatenate([s_scene(3000, B, N), rs.random_sample((B, N, 3)).astype(np.float32)], 2)).to(dev)
labels = torch.from_numpy(rs.randint(0, 9, (B, N)).astype(np.int64)).to(dev)
smpw = torch.from_numpy((rs.random_sample((B, N)) + 0.5).astype(np.float32)).to(dev)
tr = pn2.train.Trainer(hp, 9, store=pn2.util.tf_util.VariableStore(device=dev, seed=0), device=dev)
pcs = [pc, pc.clone()]
for i in range(8):
    tr.train_step(pcs[i % 2], labels, smpw, next_pc=pcs[(i + 1) % 2])
for mode in ("prefetch", "no-prefetch"):
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    K = 30
    for i in range(K):
        h0 = time.perf_counter()
        tr.train_step(pcs[i % 2], labels, smpw, next_pc=pcs[(i + 1) % 2] if mode == "prefetch" else None, sync=False)
        host.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / K
    print("%-12s step %.3f ms   host enqueue per call: median %.3f ms  max %.3f ms" % (mode, el * 1e3, np.median(host) * 1e3, max(host) * 1e3))
# graph replay alone (same inputs every time: timing only)
torch.cuda.synchronize()
with torch.cuda.stream(tr._stream):
    t0 = time.perf_counter()
    for i in range(30):
        tr._graph.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print("graph replay alone: %.3f ms per step, host launch %.3f ms per call" % ((t2 - t0) / 30 * 1e3, (t1 - t0) / 30 * 1e3))
# geometry alone
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    g = pn2.model.compute_geometry(pc[:, :, :3].contiguous(), hp, plans=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("geometry (+plans) alone: %.3f ms per batch, host %.3f ms" % ((t2 - t0) / 20 * 1e3, (t1 - t0) / 20 * 1e3))

# ---- the pieces of train_step's graph branch, timed one by one on the host (graph replays in flight) --------------------
from pn2_amd import model as M
import collections
acc = collections.OrderedDict()
def T(name, t0):
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
torch.cuda.synchronize()
K = 30
for i in range(K):
    pc_i, nxt = pcs[i % 2], pcs[(i + 1) % 2]
    t0 = time.perf_counter(); tr._lr_slot.fill_(1e-3); T("hyper host copy", t0)
    caller = torch.cuda.current_stream()
    t0 = time.perf_counter(); tr._stream.wait_stream(caller); T("wait_stream", t0)
    with torch.cuda.stream(tr._stream):
        t0 = time.perf_counter(); geo = tr._geometry_for(pc_i, tr._stream); T("_geometry_for", t0)
        t0 = time.perf_counter(); tr._lr_slot.fill_(1e-3); T("hyper H2D", t0)
        t0 = time.perf_counter(); tr.store.set_step(i); T("set_step", t0)
        t0 = time.perf_counter()
        for dst, src in zip(tr._static[:3], (pc_i, labels, smpw)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        T("input copies", t0)
        t0 = time.perf_counter(); pn2.util.tf_util.multi_copy_(M.geometry_tensors(tr._static_geo), M.geometry_tensors(geo)); T("geometry multi_copy", t0)
        t0 = time.perf_counter(); taken = torch.cuda.Event(); taken.record(tr._stream); T("event", t0)
        t0 = time.perf_counter(); tr._prefetch(nxt, taken); T("prefetch enqueue", t0)
        t0 = time.perf_counter(); tr._graph.replay(); T("graph replay", t0)
    t0 = time.perf_counter(); caller.wait_stream(tr._stream); T("caller.wait_stream", t0)
torch.cuda.synchronize()
for k, v in acc.items():
    print("  %-24s %.3f ms per step" % (k, v / K * 1e3))

# ---- where the remaining gap between "graph alone" and the step comes from ------------------------------------------------
def run(label, body, K=30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        body(i)
    torch.cuda.synchronize()
    print("  %-58s %.3f ms per step" % (label, (time.perf_counter() - t0) / K * 1e3))

geo_fixed = M.compute_geometry(pc[:, :, :3].contiguous(), hp, plans=True)
def replay_only(i):
    with torch.cuda.stream(tr._stream):
        tr._graph.replay()
def replay_copies(i):
    with torch.cuda.stream(tr._stream):
        tr._lr_slot.fill_(1e-3); tr.store.set_step(i)
        tr._static[0].copy_(pcs[i % 2], non_blocking=True)
        pn2.util.tf_util.multi_copy_(M.geometry_tensors(tr._static_geo), M.geometry_tensors(geo_fixed))
        tr._graph.replay()
def replay_side_geometry_independent(i):
    with torch.cuda.stream(tr._geo_stream):
        M.compute_geometry(pcs[i % 2][:, :, :3].contiguous(), hp, plans=True)
    with torch.cuda.stream(tr._stream):
        tr._graph.replay()
def replay_side_geometry_noplans(i):
    with torch.cuda.stream(tr._geo_stream):
        M.compute_geometry(pcs[i % 2][:, :, :3].contiguous(), hp, plans=False)
    with torch.cuda.stream(tr._stream):
        tr._graph.replay()
def replay_side_fps_only(i):
    with torch.cuda.stream(tr._geo_stream):
        pn2.util.pointnet_util.sa_geometry(pcs[i % 2][:, :, :3].contiguous(), 1024, 0.1, 32)
    with torch.cuda.stream(tr._stream):
        tr._graph.replay()
run("graph replay only", replay_only)
run("+ per-step copies (hyper, step, input, geometry)", replay_copies)
run("graph + independent geometry (+plans) on the side stream", replay_side_geometry_independent)
run("graph + independent geometry (no plans) on the side stream", replay_side_geometry_noplans)
run("graph + SA1 FPS/ball query only on the side stream", replay_side_fps_only)

# ---- GPU timeline of the real loop: graph duration and the gap between consecutive graphs ---------------------------------
orig_replay = tr._graph.replay
marks = []
def timed_replay():
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(torch.cuda.current_stream()); orig_replay(); b.record(torch.cuda.current_stream())
    marks.append((a, b))
class G:  # stand-in exposing replay()
    def __init__(self, g): self.g = g
    def replay(self): timed_replay()
real = tr._graph
tr._graph = G(real)
torch.cuda.synchronize()
for i in range(30):
    tr.train_step(pcs[i % 2], labels, smpw, next_pc=pcs[(i + 1) % 2], sync=False)
torch.cuda.synchronize()
tr._graph = real
dur = [a.elapsed_time(b) for a, b in marks[5:]]
gap = [marks[k][1].elapsed_time(marks[k + 1][0]) for k in range(5, len(marks) - 1)]
print("  real loop: graph duration %.3f ms (min %.3f max %.3f), gap between graphs %.3f ms (min %.3f max %.3f)"
      % (np.mean(dur), min(dur), max(dur), np.mean(gap), min(gap), max(gap)))

# ---- which dependency makes the gap: the real loop with pieces removed ----------------------------------------------------
def loop(label, next_of, ctx=None, K=30):
    global marks
    marks = []
    tr._graph = G(real)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if ctx is None:
        for i in range(K):
            tr.train_step(pcs[i % 2], labels, smpw, next_pc=next_of(i), sync=False)
    else:
        with ctx:
            for i in range(K):
                tr.train_step(pcs[i % 2], labels, smpw, next_pc=next_of(i), sync=False)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / K * 1e3
    tr._graph = real
    dur = [a.elapsed_time(b) for a, b in marks[5:]]
    gap = [marks[k][1].elapsed_time(marks[k + 1][0]) for k in range(5, len(marks) - 1)]
    print("  %-44s step %.3f  graph %.3f  gap %.3f ms" % (label, el, np.mean(dur), np.mean(gap)))

loop("real loop", lambda i: pcs[(i + 1) % 2])
loop("called from the trainer's own stream", lambda i: pcs[(i + 1) % 2], ctx=torch.cuda.stream(tr._stream))
orig_gf = tr._geometry_for
tr._geometry_for = lambda pc_, caller: geo_fixed
loop("fixed geometry, prefetch still running", lambda i: pcs[(i + 1) % 2])
loop("fixed geometry, no prefetch", lambda i: None)
tr._geometry_for = orig_gf
