"""Stage anatomy of the fused three-layer FP4 kernel (pn2_fp_mlp_fused_pre) from cycle stamps written by a tuning build
(build.py --tuning, PN2_HIP_LIBRARY=.../libpn2_tune.so): workgroups 0-3, waves 0 and 4, first tiles.
usage: PN2_HIP_LIBRARY=$PWD/open3d-pointnet2-semantic3d_amd/libpn2_tune.so python tools/chain_stage_ab.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pn2_amd as pn2
dev = torch.device("cuda:0")
L = ctypes.CDLL(os.environ["PN2_HIP_LIBRARY"])
stats = torch.zeros(4 * 2 * 4 * 8 + 2048, dtype=torch.int64, device=dev)
stats = torch.zeros(4 * 2 * 4 * 8 + 4096, dtype=torch.int64, device=dev)
if len(sys.argv) > 1:
    assert L.pn2_debug_set(13, int(sys.argv[1])) == 0   # stagger the two waves of a SIMD on / off
if len(sys.argv) > 2:
    assert L.pn2_debug_set(7, int(sys.argv[2])) == 0    # 4: four-wave workgroups (one wave per SIMD)
tfu, pu = pn2.util.tf_util, pn2.util.pointnet_util
B, N, M = 16, 8192, 1024
rs = np.random.RandomState(0)
xyz1 = torch.from_numpy(rs.rand(B, N, 3).astype(np.float32)).to(dev)
xyz2 = xyz1[:, :M].contiguous()
p1 = torch.from_numpy(rs.rand(B, N, 3).astype(np.float32)).to(dev)
p2 = torch.from_numpy(rs.randn(B, M, 128).astype(np.float32)).to(dev)
dist, idx = pn2.three_nn(xyz1, xyz2)
SCHED = int(os.environ.get("PN2_CHAIN_SCHEDULE", "-1"))  # -1: the layer API's own path; 0 / 1 / 2: pn2_fp_mlp_fused_pre_schedule
ws_, bs_, c_ = [], [], 131
for w_ in (128, 128, 128):
    ws_.append(torch.from_numpy((rs.randn(c_, w_) / np.sqrt(c_)).astype(np.float32)).to(dev))
    bs_.append(torch.from_numpy((rs.randn(w_) * 0.1).astype(np.float32)).to(dev))
    c_ = w_
tfu.set_default_store(tfu.VariableStore(device=dev, seed=1))
with tfu.variable_scope("fp"):
    if SCHED >= 0:
        f = lambda: tfu.hip_fp_mlp_fused_pre(dist, idx, p1, p2, ws_, bs_, schedule=SCHED)
    else:
        f = lambda: pu.fp_features_inference(dist, idx, p1, p2, [128, 128, 128])
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    L.pn2_debug_set_chain_stats(ctypes.c_void_p(stats.data_ptr()))
    f()
    torch.cuda.synchronize()
    L.pn2_debug_set_chain_stats(ctypes.c_void_p(0))
    pn2._lib.lib.trace = []
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    tr, pn2._lib.lib.trace = pn2._lib.lib.trace, None
    agg = {}
    for name, args, s_, e_ in tr:
        agg.setdefault(name, []).append(s_.elapsed_time(e_) * 1e3)
    print("  ".join("%s %.1f us" % (k, sum(v) / len(v)) for k, v in agg.items()))
allst = stats.cpu().numpy()
st = allst[:256].reshape(4, 2, 4, 8)
tot = allst[256:256 + 1024]
if tot.any():
    hw = allst[1280:2304]
    r1, r0 = allst[2304:3328], allst[3328:4352]
    print('realtime (100 MHz ticks): first start -> last start %d, first start -> last end %d, per-wave duration min %d max %d' % (r0.max() - r0.min(), r1.max() - r0.min(), (r1 - r0).min(), (r1 - r0).max()))
    print('whole-kernel cycles per WAVE: min %d median %d max %d' % (tot[tot > 0].min(), np.median(tot[tot > 0]), tot.max()))
    cu = ((hw >> 32) << 16) | (hw & 0xffffffff & 0x0f00) | ((hw >> 13) & 0x7) << 4  # xcc, cu_id(bits 8..11), se_id(13..15)
    wg_cu = cu.reshape(256, 4)[:, 0]
    print('distinct (xcc, se, cu) hosting a workgroup:', len(set(wg_cu.tolist())), 'of 256 workgroups')
    slow = np.argsort(-tot)[:8]
    print('slowest waves (wg, wave, cycles):', [(int(i // 4), int(i % 4), int(tot[i])) for i in slow])
names = ["start", "staged", "tile begin", "layer-1 input ready", "layer 2 done", "layer 3 done", "stored"]
for blk in range(4):
    for w in range(2):
        t0 = st[blk, w, 0, 0]
        row = ["wg %d wave %d: staging %6d cyc" % (blk, w * 4, st[blk, w, 0, 1] - t0)]
        for t in range(4):
            if st[blk, w, t, 2] == 0:
                continue
            v = st[blk, w, t]
            row.append("tile %d @%7d: gather+blend %6d, layer2 %6d, layer3 %6d, store %6d" % (
                t, v[2] - t0, v[3] - v[2], v[4] - v[3], v[5] - v[4], v[6] - v[5]))
        print("\n   ".join(row))
