#!/usr/bin/env python
"""A/B of pn2_linear_wgrad over the training path's layer shapes: time vs the waves-in-flight target
(pn2_debug_set(9, W), 0 = the entry point's own choice) and vs torch's x.T @ dy, plus the max abs error."""
import ctypes, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.environ.get("PN2_HIP_LIBRARY") or os.path.join(ROOT, "open3d-pointnet2-semantic3d_amd", "libpn2_hip.so"))
P = ctypes.c_void_p

SHAPES = [(524288, 9, 32), (524288, 32, 32), (524288, 32, 64), (131072, 67, 64), (131072, 64, 64), (131072, 64, 128),
          (32768, 131, 128), (32768, 128, 128), (32768, 128, 256), (8192, 259, 256), (8192, 256, 256), (8192, 256, 512),
          (1024, 768, 256), (4096, 384, 256), (16384, 320, 256), (16384, 256, 128), (131072, 134, 128),
          (131072, 128, 128), (131072, 128, 9)]

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

def main():
    dev = torch.device("cuda:0")
    st = P(torch.cuda.current_stream().cuda_stream)
    tot = {}
    for rows, cin, cout in SHAPES:
        x = torch.randn(rows, cin, device=dev); dy = torch.randn(rows, cout, device=dev)
        dw = torch.empty(cin, cout, device=dev)
        ref = (x.double().t() @ dy.double()).float()
        line = f"{rows:7d} x {cin:3d} -> {cout:3d}:"
        for w in (0, 256, 512, 768, 1536):
            lib.pn2_debug_set(9, w)
            f = lambda: lib.pn2_linear_wgrad(rows, cin, cout, P(x.data_ptr()), P(dy.data_ptr()), P(dw.data_ptr()), st)
            t = timeit(f)
            err = (dw - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)
            tot[w] = tot.get(w, 0) + t
            line += f"  W={w}: {t:7.1f} us (err {err:.1e})"
        lib.pn2_debug_set(9, 0)
        t = timeit(lambda: torch.mm(x.t(), dy))
        tot["torch"] = tot.get("torch", 0) + t
        gb = rows * (cin + cout) * 4 / 1e9
        line += f"  torch: {t:7.1f} us   [hbm floor {gb / 8e3 * 1e6:5.1f} us, mfma floor {2.0 * rows * cin * cout / 157e12 * 1e6:5.1f} us]"
        print(line, flush=True)
    print("totals (us):", {k: round(v, 1) for k, v in tot.items()})

if __name__ == "__main__":
    main()
