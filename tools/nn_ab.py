"""three_nn timing at the model's four FP shapes (HIP events, 50 reps).  usage: python tools/nn_ab.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pn2_amd as pn2
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import s_scene
dev = torch.device("cuda:0")
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
    print("three_nn(b=%d,n=%d,m=%d) %.1f us" % (b, n, m, s.elapsed_time(e) * 20))
