import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pn2_amd as pn2
from bench import s_scene, time_call
dev = torch.device("cuda:0")
import ctypes
raw = pn2._lib._raw
stats = torch.zeros(16, dtype=torch.int64, device=dev)
if hasattr(raw, "pn2_debug_set_fps_large_stats"):
    raw.pn2_debug_set_fps_large_stats(ctypes.c_void_p(stats.data_ptr()))
for (b, n, m) in [(1, 65536, 4096), (16, 65536, 4096), (1, 131072, 4096), (4, 20000, 1024)]:
    x = torch.from_numpy(s_scene(5001, b, n)[:, :, :3].copy()).to(dev)
    f = lambda: pn2.tf_ops.tf_sampling.farthest_point_sample_and_gather(m, x)
    f(); torch.cuda.synchronize()
    t = time_call(f, 3, warmup=1)
    print("fps_large b=%d n=%d m=%d: %.3f ms (%.0f ns per pick)" % (b, n, m, t, t * 1e6 / (m - 1)))
    s_ = stats.cpu().numpy()
    if s_[4]:
        ph = s_[4]
        print("   phases %d (fallback %d), %.1f picks/phase, work entries/phase %.1f, list %.1f; cycles/phase: A1 %d A2 %d fallback %d B %d" % (
            ph, s_[5], (m - 1) / ph, s_[6] / ph, s_[7] / max(1, ph - s_[5]), s_[0] // ph, s_[1] // ph, s_[2] // ph, s_[3] // ph))
