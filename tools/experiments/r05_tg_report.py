import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, pn2_amd as pn2
dev = torch.device("cuda:0")
S = pn2.tf_ops.tf_sampling
P = lambda t: ctypes.c_void_p(t.data_ptr())
L = ctypes.CDLL(os.path.join(ROOT, "tools/dbg/tg/libtg_report.so"))
for name, gen in (("scene", bench.s_scene), ("randn", bench.s_randn)):
    for b, n, m in ((16, 8192, 1024), (16, 1024, 256)):
        x = torch.from_numpy(np.ascontiguousarray(gen(7, b, n)[..., :3])).to(dev)
        _, known = S.farthest_point_sample_and_gather(m, x)
        d = torch.empty(b, n, 3, device=dev); i = torch.empty(b, n, 3, dtype=torch.int32, device=dev)
        L.tg_launch(b, n, m, P(x), 3, P(known), P(d), P(i), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        i1 = i[..., 0].cpu().numpy()
        tot = i1.size
        print(name, n, m, "not final: %.3f %%  (fewer than 3 found %.3f %%, list %.3f %%, face %.3f %%)" % (
            100.0 * (i1 < 0).mean(), 100.0 * (i1 == -2).mean(), 100.0 * (i1 == -3).mean(), 100.0 * (i1 == -4).mean()),
            "per-workgroup max:", max(int((i1.reshape(b, -1, 512 if n >= 512 else n) < 0).sum(-1).max()), 0))
