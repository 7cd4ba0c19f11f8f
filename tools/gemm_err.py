"""rounding error vs float64 of pn2_linear / pn2_linear_dgrad and torch.mm (hipBLASLt) on the model's layer shapes"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pn2_amd as pn2
tfu = pn2.util.tf_util
dev = torch.device("cuda:0")
torch.manual_seed(0)
for rows, cin, cout in [(131072, 128, 128), (1024, 768, 256), (128, 768, 256), (128, 256, 256), (256, 384, 256), (256, 256, 256), (512, 320, 256), (512, 256, 128), (16384, 131, 128), (16384, 128, 9), (65536, 6, 32), (65536, 32, 64), (16384, 67, 64), (8192, 131, 128), (4096, 259, 256), (4096, 256, 512), (2048, 64, 32), (2000, 259, 256)]:
    x = torch.randn(rows, cin, device=dev).abs() * 3 + 1; w = torch.randn(cin, cout, device=dev) / cin ** 0.5
    ref = (x.double() @ w.double())
    e_p = (tfu.hip_matmul(x, w).double() - ref).pow(2).mean().sqrt().item()
    e_t = ((x @ w).double() - ref).pow(2).mean().sqrt().item()
    dy = torch.randn(rows, cout, device=dev)
    refd = dy.double() @ w.double().t()
    d_p = (tfu.hip_linear_dgrad(dy, w).double() - refd).pow(2).mean().sqrt().item()
    d_t = ((dy @ w.t()).double() - refd).pow(2).mean().sqrt().item()
    print("%7d x %3d -> %3d  max|ref| %.1f  rms err fwd: pn2 %.2e torch %.2e | dgrad: pn2 %.2e torch %.2e" % (rows, cin, cout, ref.abs().max().item(), e_p, e_t, d_p, d_t))
