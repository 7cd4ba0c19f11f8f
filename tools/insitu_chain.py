"""In-situ duration of the FP4 chain kernel (fp_chain_pipe_kernel) WITHOUT a profiler: the tuning build stamps every wave's start
and end with the chip-wide 100 MHz clock into a ring per graph (csrc/pn2_sa_fused.hip, SaFusedParams::trace); this script runs the
throughput regime as bench.py does and reads the rings.  rocprofv3 --kernel-trace serialises the batches (the same run steps at
0.54 ms instead of 0.38), so its "in situ" kernel durations describe the profiled run, not the regime `value` is quoted in.

    python open3d-pointnet2-semantic3d_amd/build.py --tuning      (or tools/dbg/build_both.py)
    gpurun -- 'PN2_HIP_LIBRARY=$PWD/open3d-pointnet2-semantic3d_amd/libpn2_tune.so python tools/insitu_chain.py'
"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pn2_amd as pn2  # noqa: E402

raw = pn2._lib._raw
assert hasattr(raw, "pn2_debug_set_chain_trace"), "needs the tuning build (PN2_HIP_LIBRARY=.../libpn2_tune.so)"
dev = torch.device("cuda:0")
hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
B, N = 16, 8192
tfu = pn2.util.tf_util
tfu.set_default_store(tfu.VariableStore(device=dev, seed=0))
NTAG, NW, RING = 8, 1024, 16
trace = torch.zeros(NTAG * NW + NTAG * NW * RING * 2, dtype=torch.int64, device=dev)
raw.pn2_debug_set_chain_trace(ctypes.c_void_p(trace.data_ptr()))


def spans(tag_list, label):
    t = trace.cpu().numpy()
    cnt = t[:NTAG * NW].reshape(NTAG, NW)
    ring = t[NTAG * NW:].reshape(NTAG, NW, RING, 2)
    rows = []
    for tag in tag_list:
        c = cnt[tag]
        if c.max() < RING + 2 or c.min() != c.max():
            continue
        for r in range(RING):
            st, en = ring[tag, :, r, 0], ring[tag, :, r, 1]
            if (en - st).min() <= 0 or st.max() - st.min() > 100000:
                continue  # a ring slot caught between two launches
            span = (en.max() - st.min()) / 100.0          # us
            late = (st - st.min()) / 100.0
            wg_late = late.reshape(256, 4).min(axis=1)
            rows.append((span, (en - st).mean() / 100.0, int((wg_late > 20.0).sum()), wg_late.max()))
    a = np.array(rows)
    print("%-44s launches %3d: span avg %.1f us (min %.1f, median %.1f, max %.1f); mean wave %.1f us; workgroups starting "
          "> 20 us late: avg %.1f of 256 (max %d), in %.0f %% of the launches" % (
              label, len(a), a[:, 0].mean(), a[:, 0].min(), np.median(a[:, 0]), a[:, 0].max(), a[:, 1].mean(), a[:, 2].mean(),
              a[:, 2].max(), 100.0 * (a[:, 2] > 0).mean()), flush=True)
    return a


def make(n):
    raw.pn2_debug_set(16, n)  # the launches captured for this slot carry tag n
    return torch.from_numpy(bench.s_scene(2000 + n, B, N)).to(dev)


with torch.no_grad():
    pn2.model.get_sa_fp_features(make(0), False, hp)
for backlog in ((0, 0, 0, 0), (0, 0, 1, 1), (0, 0, 0, 0), (0, 0, 1, 1)):
    trace.zero_()
    pipe = pn2.runtime.StaggeredPipeline(lambda x: pn2.model.sa1_samples(x, hp),
                                         lambda x, s: pn2.model.get_sa_fp_features(x, False, hp, sa1=s)[0], make, backlog)
    for _ in range(10):
        pipe.step()
    pipe.flush()
    torch.cuda.synchronize()
    K = 240
    for rep in range(2):  # (the first timing of a process runs slow: clocks / queues still settling; the second is reported)
        trace.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            pipe.step()
        pipe.flush()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / K * 1e3
        print("   (timing %d: %.4f ms per step)" % (rep, ms))
    print("StaggeredPipeline backlog %s: %.4f ms per step over %d steps (tuning build)" % (backlog, ms, K))
    spans(range(pipe.batches_in_flight), "  FP4 chain, throughput regime (unprofiled)")
    # one batch at a time on the same graphs
    trace.zero_()
    for _ in range(40):
        with torch.cuda.stream(pipe.streams[0]):
            pipe.slots[0][0][1].replay()
            pipe.slots[0][0][2].replay()
        torch.cuda.synchronize()
    spans([0], "  FP4 chain, one batch in flight")
    del pipe
