"""The resident-weight dense kernel (csrc/pn2_linear_wres.h) against linear_kernel, same entry points, hook 17 off / on
(tuning build: PN2_HIP_LIBRARY=.../libpn2_tune.so), at the many-row layers of configs[1]: plain forward (+ max over 32 rows), the
training forward (batch norm of the layer below on load + statistics + finish), the data gradient (given dy / formed on load,
epilogue for the layer below).  Graph-timed (20 launches per replay); outputs compared bit for bit.
NEEDS tools/experiments/r06_linear_wres/ applied (dispatch.patch + the header into csrc/): in the tree as it is, hook 17 selects the
streaming narrow-layer kernel (csrc/pn2_fwd_narrow.h, tools/fwd_narrow_ab.py) and this script would compare that one.
    python tools/dbg/build_both.py && gpurun -- 'PN2_HIP_LIBRARY=$PWD/open3d-pointnet2-semantic3d_amd/libpn2_tune.so python tools/lin_wres_ab.py'"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pn2_amd as pn2  # noqa: E402
raw = pn2._lib._raw
dev = torch.device("cuda:0")
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())  # noqa: E731


def graph_time(fn, reps=20, iters=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        st = ctypes.c_void_p(s.cuda_stream)
        fn(st); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn(st)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(iters):
            e0.record(s); g.replay(); e1.record(s); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


shapes = [(131072, 128, 128, 0), (131072, 64, 128, 0), (131072, 64, 64, 0), (32768, 128, 256, 32), (32768, 128, 128, 0),
          (524288, 64, 64, 0), (131072, 256, 128, 0), (65536, 128, 128, 32), (16384, 128, 128, 0), (16384, 64, 64, 0)]
for rows, cin, cout, pool in shapes:
    torch.manual_seed(0)
    x = torch.randn(rows, cin, device=dev); w = torch.randn(cin, cout, device=dev) / cin ** 0.5
    bias = torch.randn(cout, device=dev) * 0.1
    sc, sh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1
    g, b = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
    nb = raw.pn2_bn_workspace_bytes(cout)
    dz = torch.randn(rows, cout, device=dev); coef = torch.rand(6, cout, device=dev) + 0.5
    nb2 = raw.pn2_bn_workspace_bytes(cin)
    gb, bb, mb, ib = torch.ones(cin, device=dev), torch.zeros(cin, device=dev), torch.zeros(cin, device=dev), torch.ones(cin, device=dev)
    res = {}
    for on in (0, 7):
        raw.pn2_debug_set(17, on)
        y = torch.empty(rows // max(pool, 1), cout, device=dev)
        y2 = torch.empty(rows, cout, device=dev)
        ws = torch.zeros(nb // 8, dtype=torch.float64, device=dev)
        sm, si, s2, h2 = (torch.empty(cout, device=dev) for _ in range(4))
        dx, dx2 = torch.empty(rows, cin, device=dev), torch.empty(rows, cin, device=dev)
        ws2 = torch.zeros(nb2 // 8, dtype=torch.float64, device=dev)
        cb, dgb, dbb = torch.empty(6, cin, device=dev), torch.empty(cin, device=dev), torch.empty(cin, device=dev)
        fwd = lambda st: raw.pn2_linear(rows, cin, cout, P(x), P(w), P(bias), 1, pool, P(y), st)  # noqa: E731
        def fxf(st):
            ws.zero_()
            return raw.pn2_linear_bn_stats_fin(rows, cin, cout, P(x), P(w), P(y2), P(ws), nb, P(sc), P(sh), 1, 2, P(g), P(b), None,
                                               ctypes.c_float(1e-3), ctypes.c_float(0.9), P(rm), P(rv), P(sm), P(si), P(s2), P(h2), st)
        def dgx(st):
            ws2.zero_()
            return raw.pn2_linear_dgrad_fin(rows, cin, cout, None, P(y2), P(dz), P(coef), 1, 0, None, None, P(w), P(dx), P(x), P(gb),
                                            P(bb), P(mb), P(ib), 1, P(ws2), nb2, 3, P(cb), P(dgb), P(dbb), st)
        dgp = lambda st: raw.pn2_linear_dgrad(rows, cin, cout, P(dz), P(w), P(dx2), st)  # noqa: E731
        st0 = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rcs = []
        for f_ in (fwd, fxf, dgx, dgp):
            rcs.append(f_(st0))
            torch.cuda.synchronize()
            if os.environ.get("PN2_AB_VERBOSE"):
                print("hook", on, f_.__name__, "rc", rcs[-1], flush=True)
        rcs = tuple(rcs)
        keep = [t.clone() for t in (y, y2, sm, si, dx, cb, dgb, dx2)]
        t = [graph_time(f) if rc == 0 else float("nan") for f, rc in zip((fwd, fxf, dgx, dgp), rcs)]
        res[on] = (t, keep, rcs)
    same = [bool(torch.equal(a, b_)) for a, b_ in zip(res[0][1], res[7][1])]
    close = [float((a - b_).abs().max()) for a, b_ in zip(res[0][1], res[7][1])]
    fl = 2.0 * rows * cin * cout
    print("(%d, %d -> %d, pool %d)  rc %s" % (rows, cin, cout, pool, res[7][2]))
    for i, nm in enumerate(("pn2_linear", "bn_stats_fin (xf)", "dgrad_fin (gx)", "dgrad")):
        a, b_ = res[0][0][i], res[7][0][i]
        print("   %-18s linear_kernel %6.1f us (%5.1f TF)   resident %6.1f us (%5.1f TF)" % (nm, a, fl / a * 1e-6, b_, fl / b_ * 1e-6))
    print("   bit-equal [y, y_xf, mean, invstd, dx_gx, coef_below, dgamma_below, dx]: %s  max |diff| %s" % (same, ["%.1e" % c for c in close]))
raw.pn2_debug_set(17, 7)
