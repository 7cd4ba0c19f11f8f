"""three_nn timing at the model's four FP shapes (HIP events, 50 reps).  usage: python tools/nn_ab.py [blocks ...]
With PN2_HIP_LIBRARY pointing at a tuning build (build.py --tuning) the listed workgroup targets (hook 12) are swept."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pn2_amd as pn2
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import s_scene
dev = torch.device("cuda:0")
L = pn2._lib.lib._dll if hasattr(pn2._lib.lib, "_dll") else None
sweep = [int(a) for a in sys.argv[1:]] or [0]
for blocks in sweep:
    if blocks:
        import ctypes
        dll = ctypes.CDLL(os.environ["PN2_HIP_LIBRARY"])
        assert dll.pn2_debug_set(12, blocks) == 0
    row = ["blocks=%d" % blocks]
    for n, m in [(8192, 1024), (1024, 256), (256, 64), (64, 16), (65536, 4096)]:
        b = 16 if n < 65536 else 1
        a = torch.from_numpy(s_scene(1, b, n)).to(dev)
        r = a[:, :m].contiguous()
        for _ in range(5):
            pn2.three_nn(a, r)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(50):
            pn2.three_nn(a, r)
        e.record(); torch.cuda.synchronize()
        row.append("(%d,%d,%d) %.1f us" % (b, n, m, s.elapsed_time(e) * 20))
    print("  ".join(row), flush=True)
