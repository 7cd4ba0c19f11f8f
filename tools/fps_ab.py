"""A/B timing of libpn2_hip.so variants (direct ctypes, no package): FPS / ball query / three_nn."""
import ctypes, sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def scene(seed, b, n):
    rs = np.random.RandomState(seed)
    return np.concatenate([rs.uniform(-5, 5, (b, n, 2)), np.clip(np.abs(rs.normal(0, 1.5, (b, n, 1))), 0, 8)], 2).astype(np.float32)

def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

def main():
    libs = [a for a in sys.argv[1:] if not a.startswith("--")]
    variants = [int(a[2:]) for a in sys.argv[1:] if a.startswith("--")] or [0]
    dev = torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    ref = {}
    for path, var in [(p, v) for p in libs for v in variants]:
        L = ctypes.CDLL(os.path.abspath(path))
        L.pn2_debug_set(0, var)
        stats = torch.zeros(64, dtype=torch.int64, device=dev)
        if hasattr(L, "pn2_debug_set_fps_stats"):
            L.pn2_debug_set_fps_stats(ctypes.c_void_p(stats.data_ptr()))
        row = [os.path.basename(path) + ":%d" % var]
        for (b, n, m) in [(16, 8192, 1024), (16, 1024, 256), (16, 256, 64), (16, 64, 16), (16, 4096, 512), (16, 2048, 256)]:
            x = torch.from_numpy(scene(n, b, n)).to(dev)
            out = torch.empty((b, m), dtype=torch.int32, device=dev)
            f = lambda: L.pn2_farthest_point_sample(b, n, m, P(x), None, P(out), 1, st)
            assert f() == 0
            t = timeit(f)
            key = (b, n, m)
            o = out.cpu().numpy()
            if key in ref: assert (ref[key] == o).all(), "variant disagrees at %s" % (key,)
            ref[key] = o
            row.append("fps%s=%.1fus(%.3fus/round)" % (key, t, t / (m - 1)))
            s_ = stats.cpu().numpy()
            if s_[0] > 0:
                tot = max(1, s_[8])
                ph = s_[0]
                row.append("\n    [lazy: %d phases (%d empty, %d overflow), mean list %.1f, %.1f picks/phase; wave0 cycles: A %.0f%% wait1 %.0f%% B %.0f%% "
                           "wait2 %.0f%% of %d (%.0f cyc/phase, B %.0f cyc/pick)\n     per wave cyc/phase: A %s\n       of which bbox+update %s\n       pairs/phase %s]\n   " % (
                    s_[0], s_[1], s_[2], s_[3] / max(1, s_[0] - s_[1] - s_[2]), (m - 1) / s_[0], 100 * s_[4] / tot, 100 * s_[5] / tot,
                    100 * s_[6] / tot, 100 * s_[7] / tot, tot, tot / s_[0], s_[6] / (m - 1), (s_[16:32] // ph).tolist(), (s_[48:64] // ph).tolist(),
                    [round(float(v) / ph, 1) for v in s_[32:48]]))
                stats.zero_()
        print("  ".join(row))

if __name__ == "__main__":
    main()
