"""pn2_linear tile configurations (tuning hook 8) at the few-row shapes of the hoisted products / FP1:
PN2_HIP_LIBRARY=.../libpn2_tune.so python tools/dbg/lin_cfg_sweep.py"""
import sys, os, ctypes
sys.path.insert(0, "/root/repo")
import torch
import pn2_amd as pn2
dev = torch.device("cuda:0")
tfu = pn2.util.tf_util
L = ctypes.CDLL(os.environ["PN2_HIP_LIBRARY"])
names = {0: "auto", 1: "<4,1,4>", 2: "<2,2,2>", 3: "<1,4,1>", 4: "<1,2,1,k2>", 5: "splitk g2", 6: "splitk g4"}
for rows, cin, cout in [(16384, 128, 128), (4096, 128, 128), (4096, 256, 256), (1024, 256, 256), (1024, 768, 256), (32768, 128, 256)]:
    x = torch.randn(rows, cin, device=dev); w = torch.randn(cin, cout, device=dev)
    out = []
    for cfg in (0, 2, 3, 4, 5, 6):
        L.pn2_debug_set(8, cfg)
        f = lambda: tfu.hip_linear(x, w, None, relu=False)
        for _ in range(3): f()
        g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            f()
            with torch.cuda.graph(g, stream=s):
                for _ in range(20): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay(); torch.cuda.synchronize()
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        out.append("%s %.2f" % (names[cfg], a.elapsed_time(b) * 50))
    L.pn2_debug_set(8, 0)
    print((rows, cin, cout), "  ".join(out))
