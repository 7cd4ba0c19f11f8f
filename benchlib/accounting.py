"""What a launch is worth: roofline constants, algorithmic bytes / flops per launch of every entry point,
the summary of the HIP-event trace of an instrumented pass, the committed PMC traffic of the roofline kernel."""
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TF = 157.3   # dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)

FPS_KERNELS = ("farthest_point_sample", "fps_gather", "fps_nested")
# one workgroup per cloud, latency-bound like the samplers (they hide behind the other batches when several are in flight)
LATENCY_KERNELS = FPS_KERNELS + ("coarse_geometry", "ball_query_bin")

# HBM bytes per launch of the roofline kernel: PMC counters are a separate rocprofv3 pass (--pmc FETCH_SIZE / WRITE_SIZE cannot
# share a run with the timed region), so the line cites the committed summary of that pass (tools/gpu_round4.sh,
# tools/pmc_to_profiles.py: gfx950 corrections as MI355X_MICROARCH.md prescribes) instead of carrying `null`.
PMC_FILE = "profiles/r06_pmc_hbm_traffic.json"
PMC_KERNEL_OF = {"fp_mlp_fused_pre": "fp_chain_pipe_kernel", "fp_mlp_fused": "sa_fused_kernel<2", "sa_mlp_max_fused": "sa_fused_kernel<3, 1, 1, 2"}


def pmc_traffic(kernel):
    """bytes per launch of `kernel` from the committed PMC summary, or None (file absent / kernel not in it)"""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            ks = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    key = PMC_KERNEL_OF.get(kernel, kernel)
    hits = [v["traffic_bytes"] for k, v in ks.items() if key in k]
    return max(hits) if hits else None


# ---- algorithmic work per launch (compulsory traffic) -----------------
def kernel_model(name, a):
    """-> (bound, units) with units = algorithmic bytes (hbm) or flops (mfma) of ONE launch."""
    if name in ("pn2_farthest_point_sample", "pn2_fps_gather", "pn2_fps_nested"):
        b, n, m = a[0], a[1], a[2]
        return "hbm", b * n * 12 + b * m * 4 + (b * m * 12 if name != "pn2_farthest_point_sample" else 0)
    if name == "pn2_coarse_geometry":  # ints: b, n0, nlev, fps mode, bq mode, then (decoded by _lib) npoint[], nsample[]
        b, n, L = a[:3]
        byts = b * n * 12
        for m, ns in zip(a[5:5 + L], a[5 + L:5 + 2 * L]):
            byts += b * (m * 16 + m * ns * 4 + m * 4 + n * 24)
            n = m
        return "hbm", byts
    if name == "pn2_gather_point":
        b, n, m = a[:3]
        return "hbm", b * m * 4 + b * m * 12 * 2
    if name in ("pn2_query_ball_point", "pn2_query_ball_point_binned"):
        b, n, m, _, ns = a[:5]
        return "hbm", b * n * 12 + b * m * 12 + b * m * ns * 4 + b * m * 4
    if name == "pn2_ball_query_bin":
        b, n = a[:2]
        return "hbm", b * n * 12 + b * n * 14
    if name == "pn2_group_point":
        b, n, c, m, ns = a[:5]
        return "hbm", b * m * ns * 4 + b * n * c * 4 + b * m * ns * c * 4
    if name == "pn2_sa_group_concat":
        b, n, m, ns, c = a[:5]
        return "hbm", b * m * ns * 4 + b * n * (3 + c) * 4 + b * m * 12 + b * m * ns * (3 + c) * 4
    if name == "pn2_three_nn":
        b, n, m = a[:3]
        return "hbm", b * n * 12 + b * m * 12 + b * n * 24
    if name == "pn2_three_interpolate":
        b, m, c, n = a[:4]
        return "hbm", b * m * c * 4 + b * n * 24 + b * n * c * 4
    if name == "pn2_fp_interp_concat":
        b, n, m, c1, c2 = a[:5]
        return "hbm", b * m * c2 * 4 + b * n * 24 + b * n * c1 * 4 + b * n * (c1 + c2) * 4
    if name == "pn2_linear":
        rows, cin, cout = a[:3]
        return "mfma", 2 * rows * cin * cout
    if name == "pn2_mlp_chain":
        rows, cin, L = a[0], a[1], a[2]
        widths = a[4:4 + L]
        fl = 0
        for w in widths:
            fl += 2 * rows * cin * w
            cin = w
        return "mfma", fl
    if name == "pn2_mlp_wide":      # ints: rows, cin, x_stride, nlayers, relu_last, pool, widths...
        rows, cin, L = a[0], a[1], a[3]
        fl = 0
        for w in a[6:6 + L]:
            fl += 2 * rows * cin * w
            cin = w
        return "mfma", fl
    if name == "pn2_sa_mlp_wide":   # ints: b, n, m, nsample, c, nlayers, pool, widths...
        b, n, m, ns, c, L = a[:6]
        cin, fl = 3 + c, 0
        for w in a[7:7 + L]:
            fl += 2 * b * m * ns * cin * w
            cin = w
        return "mfma", fl
    if name == "pn2_fp_mlp_wide":   # ints: b, n, m, c1, c2, nlayers, widths...
        b, n, m, c1, c2, L = a[:6]
        cin, fl = c1 + c2, 0
        for w in a[6:6 + L]:
            fl += 2 * b * n * cin * w
            cin = w
        return "mfma", fl
    if name == "pn2_fp_mlp_fused":
        b, n, m, c1, c2, L = a[:6]
        widths = a[6:6 + L]
        cin, fl = c1 + c2, 0
        for w in widths:
            fl += 2 * b * n * cin * w
            cin = w
        return "mfma", fl
    if name == "pn2_fp_mlp_wide_pre":  # ints: b, n, m, c1, nlayers, widths...   EXECUTED flops
        b, n, m, c1, L = a[:5]
        widths = a[5:5 + L]
        fl, cin = 2 * b * n * c1 * widths[0], widths[0]
        for w in widths[1:]:
            fl += 2 * b * n * cin * w
            cin = w
        return "mfma", fl
    if name == "pn2_sa_mlp_wide_pre":  # ints: b, n, m, nsample, nlayers, pool, widths...
        b, n, m, ns, L = a[:5]
        widths = a[6:6 + L]
        fl, cin = 2 * b * m * ns * 3 * widths[0], widths[0]
        for w in widths[1:]:
            fl += 2 * b * m * ns * cin * w
            cin = w
        return "mfma", fl
    if name == "pn2_sa_mlp_fused_pre":  # ints: b, n, m, nsample, nlayers, pool, widths...  EXECUTED flops (xyz rows + later layers)
        b, n, m, ns, L = a[:5]
        widths = a[6:6 + L]
        fl = 2 * b * m * ns * 3 * widths[0]
        cin = widths[0]
        for w in widths[1:]:
            fl += 2 * b * m * ns * cin * w
            cin = w
        return "mfma", fl
    if name == "pn2_fp_mlp_fused_pre":  # ints: b, n, m, c1, nlayers, widths...  EXECUTED flops: skip channels + later layers
        b, n, m, c1, L = a[:5]
        widths = a[5:5 + L]
        fl = 2 * b * n * c1 * widths[0]
        cin = widths[0]
        for w in widths[1:]:
            fl += 2 * b * n * cin * w
            cin = w
        return "mfma", fl
    if name in ("pn2_sa_mlp_max_fused", "pn2_sa_mlp_rows_fused"):
        b, n, m, ns, c, L = a[:6]
        widths = a[6:6 + L]
        cin, fl = 3 + c, 0
        for w in widths:
            fl += 2 * b * m * ns * cin * w
            cin = w
        return "mfma", fl
    return "hbm", 0


def reference_flops(pn2, hp, B, N):
    """MFMA flops of one step in the reference's own formulation (every 1x1 conv of every SA / FP module applied to the
    grouped / concatenated tensor: SURVEY.md 8(d)'s 2*rows*cin*cout).  The product executes fewer: the first layer of a
    module is applied to the SOURCE rows where linearity allows (DESIGN.md 4 'hoisting')."""
    mdl = pn2.model
    npts = [N] + [hp["l%d_npoint" % i] for i in (1, 2, 3, 4)]
    width = [3 * int(hp["use_color"])] + [w[-1] for w in mdl.SA_MLPS]
    fl = 0
    for li in range(4):
        rows, cin = B * npts[li + 1] * hp["l%d_nsample" % (li + 1)], 3 + width[li]
        for w in mdl.SA_MLPS[li]:
            fl, cin = fl + 2 * rows * cin * w, w
    up = width[4]
    for fi in range(4):
        lvl = 3 - fi
        rows, cin = B * npts[lvl], width[lvl] + up
        for w in mdl.FP_MLPS[fi]:
            fl, cin = fl + 2 * rows * cin * w, w
        up = cin
    return fl


def summarize_trace(trace, steps):
    """aggregate (name, args) -> avg ms per launch, launches per step, roofline numbers."""
    agg = {}
    for name, args, s, e in trace:
        key = (name, args)
        d = agg.setdefault(key, [0.0, 0])
        d[0] += s.elapsed_time(e)
        d[1] += 1
    rows = []
    for (name, args), (tot_ms, cnt) in agg.items():
        bound, units = kernel_model(name, args)
        avg_ms = tot_ms / cnt
        if bound == "hbm":
            ach = units / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            peak, unit = HBM_PEAK_GBS, "GB/s"
        else:
            ach = units / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
            peak, unit = MFMA_F32_PEAK_TF, "TFLOP/s"
        rows.append({"kernel": name.replace("pn2_", ""), "args": list(args), "bound": bound,
                     "avg_us": round(avg_ms * 1e3, 2), "launches_per_step": cnt / steps,
                     "ms_per_step": round(tot_ms / steps, 4), "achieved": round(ach, 3), "peak": peak,
                     "unit": unit, "frac": round(ach / peak, 5),
                     "algorithmic_units": int(units)})
    rows.sort(key=lambda r: -r["ms_per_step"])
    return rows


def time_call(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters  # ms
