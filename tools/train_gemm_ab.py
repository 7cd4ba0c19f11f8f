#!/usr/bin/env python
"""A/B for the training path's two dense GEMMs per layer: y = x @ w (forward) and dx = dy @ w^T, torch.mm
(hipBLASLt) against pn2_linear (fp32 MFMA kernel of the inference path, bias / ReLU off)."""
import ctypes, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "open3d-pointnet2-semantic3d_amd", "libpn2_hip.so"))
P = ctypes.c_void_p

# (rows, cin, cout) of every dense layer of the semantic.json model at B=16, N=8192
LAYERS = [(524288, 6, 32), (524288, 32, 32), (524288, 32, 64), (131072, 67, 64), (131072, 64, 64), (131072, 64, 128),
          (32768, 131, 128), (32768, 128, 128), (32768, 128, 256), (8192, 259, 256), (8192, 256, 256), (8192, 256, 512),
          (1024, 768, 256), (1024, 256, 256), (4096, 384, 256), (4096, 256, 256), (16384, 320, 256), (16384, 256, 128),
          (131072, 134, 128), (131072, 128, 128), (131072, 128, 128), (131072, 128, 128)]

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

def main():
    dev = torch.device("cuda:0")
    st = P(torch.cuda.current_stream().cuda_stream)
    tot = {"fwd_torch": 0, "fwd_pn2": 0, "dx_torch": 0, "dx_pn2": 0}
    for rows, cin, cout in LAYERS:
        x = torch.randn(rows, cin, device=dev); w = torch.randn(cin, cout, device=dev) * 0.1
        dy = torch.randn(rows, cout, device=dev); wt = w.t().contiguous()
        y = torch.empty(rows, cout, device=dev); dx = torch.empty(rows, cin, device=dev)
        t_f = timeit(lambda: torch.mm(x, w, out=y))
        rc = lib.pn2_linear(rows, cin, cout, P(x.data_ptr()), P(w.data_ptr()), None, 0, 0, P(y.data_ptr()), st)
        t_fp = timeit(lambda: lib.pn2_linear(rows, cin, cout, P(x.data_ptr()), P(w.data_ptr()), None, 0, 0, P(y.data_ptr()), st)) if rc == 0 else float("nan")
        err = (y - x @ w).abs().max().item() if rc == 0 else float("nan")
        t_d = timeit(lambda: torch.mm(dy, wt, out=dx))
        t_dp = timeit(lambda: lib.pn2_linear_dgrad(rows, cin, cout, P(dy.data_ptr()), P(w.data_ptr()), P(dx.data_ptr()), st))
        derr = (dx - dy @ wt).abs().max().item()
        print(f"{rows:7d} x {cin:3d} -> {cout:3d}: fwd torch {t_f:6.1f} us  pn2_linear {t_fp:6.1f} us (err {err:.1e}) | dx torch {t_d:6.1f} us  pn2_linear_dgrad {t_dp:6.1f} us (err {derr:.1e})", flush=True)
        tot["fwd_torch"] += t_f; tot["fwd_pn2"] += t_fp if t_fp == t_fp else t_f
        tot["dx_torch"] += t_d; tot["dx_pn2"] += t_dp
    print("totals (us):", {k: round(v, 1) for k, v in tot.items()})

if __name__ == "__main__":
    main()
