"""ball query at the small SA levels: automatic kernel choice vs the LDS-grid kernel forced (kernel=3) and the lane kernel (2)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pn2_amd as pn2
from bench import s_scene


def timeit(fn, iters=20):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(iters): fn()
        g.replay(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(st); g.replay(); g.replay(); g.replay(); e.record(st); torch.cuda.synchronize()
    return s.elapsed_time(e) / (3 * iters) * 1e3


dev = torch.device("cuda:0")
hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
pc = torch.from_numpy(s_scene(3000, 16, 8192)).to(dev)
geo = pn2.model.compute_geometry(pc[:, :, :3].contiguous(), hp)
for li in range(4):
    xyz, new_xyz = geo["xyzs"][li], geo["xyzs"][li + 1]
    k = "l%d_" % (li + 1)
    r, ns = hp[k + "radius"], hp[k + "nsample"]
    ref = None
    row = ["level %d n=%d m=%d r=%.2f" % (li + 1, xyz.shape[1], new_xyz.shape[1], r)]
    for kern in (0, 2, 3):
        try:
            f = lambda: pn2.query_ball_point(r, ns, xyz, new_xyz, kernel=kern)
            idx, cnt = f()
            if ref is None: ref = idx
            assert torch.equal(idx, ref)
            row.append("kernel %d: %.1f us" % (kern, timeit(f)))
        except Exception as ex:
            row.append("kernel %d: %s" % (kern, str(ex)[:40]))
    print("  ".join(row))
