"""A/B of the SA1 sampler: the previous round's kernel (tools/ab/libfps_old.so, built from git history) against
pn2_fps_gather (no tie record) and pn2_fps_nested (tie record written) of the current library."""
import ctypes
import sys
import os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import pn2_amd as pn2  # noqa: E402
from conftest import s_scene, s_randn, s_dup  # noqa: E402

dev = torch.device("cuda:0")
cur = pn2._lib.lib
old = None
p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", "libfps_old.so")
if os.path.exists(p):
    old = ctypes.CDLL(p)
vp = ctypes.c_void_p


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1000.0


for name, gen in [("scene", s_scene), ("randn", s_randn), ("dup25", s_dup)]:
    for n, m in [(8192, 1024), (4096, 1024), (1024, 256), (256, 64)]:
        x = torch.from_numpy(gen(0, 16, n)[..., :3].copy()).to(dev)
        out = torch.empty((16, m), dtype=torch.int32, device=dev)
        nx = torch.empty((16, m, 3), dtype=torch.float32, device=dev)
        tie = torch.empty((16,), dtype=torch.int32, device=dev)
        st = vp(torch.cuda.current_stream().cuda_stream)
        args = (16, n, m, vp(x.data_ptr()), None, vp(out.data_ptr()), vp(nx.data_ptr()))
        r = {}
        if old is not None:
            r["old"] = timeit(lambda: old.pn2_fps_gather(*args, 2, st))
        r["gather"] = timeit(lambda: cur.pn2_fps_gather(*args, 2, st))
        r["nested_track"] = timeit(lambda: cur.pn2_fps_nested(*args, None, vp(tie.data_ptr()), 2, st))
        print(name, n, m, {k: round(v, 1) for k, v in r.items()}, "tie min", int(tie.min()))
