"""Do CU-masked HIP streams (hipExtStreamCreateWithCUMask) confine (a) eager launches, (b) replays of a graph captured on them,
(c) replays of a graph captured elsewhere?  And which hardware CUs do the mask bits select?  (r05: can the FPS chains of the
batches in flight be kept off the CUs the persistent MFMA grids are sized for.)

    gpurun -- 'python tools/cu_mask_probe.py'
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pn2_amd as pn2  # noqa: E402,F401
sys.path.insert(0, os.path.join(ROOT, "tools", "experiments"))
from r05_sampler_ahead import masked_stream  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", "libcu_probe.so"))
lib.cu_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
dev = torch.device("cuda:0")


def launch(out, grid, stream, spin=3000, lds=0):
    rc = lib.cu_probe_launch(out.data_ptr(), grid, 64, lds, spin, stream.cuda_stream)
    assert rc == 0, rc


def cus(out):
    a = out.cpu().numpy().astype("uint32").reshape(-1, 2)
    hw, xcc = a[:, 0], a[:, 1] & 0xF
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    return sorted(set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist())))


def describe(tag, out):
    s = cus(out)
    per_xcc = {}
    for x, se, sh, cu in s:
        per_xcc.setdefault(x, []).append((se, cu))
    print("%-44s %3d distinct CUs; per XCC: %s" % (tag, len(s), {k: len(v) for k, v in sorted(per_xcc.items())}))
    return s


G = 4096
out = torch.zeros((G, 2), dtype=torch.int32, device=dev)
plain = torch.cuda.Stream()
launch(out, G, plain); torch.cuda.synchronize()
allc = describe("plain stream, eager", out)

for nbits, label in ((16, "low 16 bits"), (32, "low 32 bits"), (8, "low 8 bits")):
    ms = masked_stream(dev, (1 << nbits) - 1)
    out.zero_(); torch.cuda.synchronize()
    launch(out, G, ms); torch.cuda.synchronize()
    s = describe("masked stream (%s), eager" % label, out)
    if nbits == 16:
        print("   CUs selected by the low 16 bits:", s)

low16 = masked_stream(dev, (1 << 16) - 1)
high240 = masked_stream(dev, ((1 << 256) - 1) ^ ((1 << 16) - 1))
out.zero_(); torch.cuda.synchronize()
launch(out, G, high240); torch.cuda.synchronize()
s240 = describe("masked stream (bits 16..255), eager", out)

# (b) graph captured ON the masked stream, replayed on it
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=low16):
    launch(out, G, torch.cuda.current_stream())
out.zero_(); torch.cuda.synchronize()
with torch.cuda.stream(low16):
    g.replay()
torch.cuda.synchronize()
describe("graph captured + replayed on low16", out)

# (c) graph captured on an ordinary stream (torch's capture stream), replayed on the masked stream
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    launch(out, G, torch.cuda.current_stream())
out.zero_(); torch.cuda.synchronize()
with torch.cuda.stream(low16):
    g2.replay()
torch.cuda.synchronize()
describe("graph captured elsewhere, replayed on low16", out)
out.zero_(); torch.cuda.synchronize()
with torch.cuda.stream(high240):
    g2.replay()
torch.cuda.synchronize()
describe("graph captured elsewhere, replayed on 16..255", out)
out.zero_(); torch.cuda.synchronize()
with torch.cuda.stream(plain):
    g2.replay()
torch.cuda.synchronize()
describe("graph captured elsewhere, replayed on plain", out)
