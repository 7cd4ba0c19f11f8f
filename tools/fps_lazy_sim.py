"""Simulation behind the lazy multi-pick FPS kernel (csrc/pn2_sampling.hip fps_lazy_kernel).

A PHASE = one full pass that applies the pending picks to every point and lists the points whose running
minimum td is >= tau (the candidate list), then a single wave picks from the list for as long as the best
candidate's td stays >= tau (every unlisted point has td < tau, so the listed maximum IS the global one).
Statistics only (ties broken by index order, not by the reference's (k mod 512, k) rule):
  picks per phase, list sizes, overflow rate for a capacity, and -- with the cloud Morton-sorted into buckets --
  how many (bucket, pending pick) pairs a phase really has to evaluate (a pick p can lower td inside a bucket
  only if lb(p, bbox)^2 < max td of the bucket).
"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import s_scene


def morton_order(x, bits=10):
    lo, hi = x.min(0), x.max(0)
    q = np.minimum(((x - lo) / np.maximum(hi - lo, 1e-9) * (1 << bits)).astype(np.int64), (1 << bits) - 1)
    code = np.zeros(len(x), np.int64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return np.argsort(code, kind="stable")


def sim(x, m, eps0, cap, bucket=None, adapt=True, target=(12, 40), global_bound=False, interleave=False):
    n = len(x)
    if bucket:
        x = x[morton_order(x)]
        nb = n // bucket
        bb_lo = x.reshape(nb, bucket, 3).min(1)
        bb_hi = x.reshape(nb, bucket, 3).max(1)
    td = np.full(n, 1e38, np.float32)
    picks = [0]
    pending = [0]
    eps = eps0
    d_last = np.float32(1e38)
    phases = 0
    hist_picks, hist_list, overflow, empty = [], [], 0, 0
    pair_work, pair_all, maxwork = 0, 0, []
    while len(picks) < m:
        # phase A: apply pending picks
        if bucket:
            tdb = td.reshape(nb, bucket)
            bmax = tdb.max(1)
            work = np.zeros(nb, int)
            gbound = td.max()  # = td of the first pending pick when it was picked (all stale td <= it)
            for p in pending:
                d = np.maximum(np.maximum(bb_lo - x[p], x[p] - bb_hi), 0)
                lb = (d * d).sum(1)
                touch = lb * (1 - 1e-6) <= (gbound if global_bound else bmax)
                work += touch
            pair_work += work.sum()
            pair_all += nb * len(pending)
            rows = max(1, 512 // bucket)
            if interleave:   # bucket b -> wave b % nwaves
                maxwork.append(work.reshape(rows, -1).sum(0).max())
            else:
                maxwork.append(work.reshape(-1, rows).sum(1).max())
        for p in pending:
            td = np.minimum(td, ((x - x[p]) ** 2).sum(1).astype(np.float32))
        phases += 1
        tau = np.float32(d_last * (1 - eps)) if d_last < 1e37 else np.float32(0.5) * td.max()
        lst = np.nonzero(td >= tau)[0]
        cnt = len(lst)
        hist_list.append(cnt)
        pending = []
        if cnt == 0 or cnt > cap:
            if cnt == 0:
                empty += 1
            else:
                overflow += 1
            k = int(np.argmax(td))
            d_last = td[k]
            picks.append(k)
            pending.append(k)
            if adapt:
                eps = eps * 0.5 if cnt > cap else min(0.9, eps * 2)
            hist_picks.append(1)
            continue
        cx = x[lst]
        ctd = td[lst].copy()
        np_ = 0
        while len(picks) < m:
            i = int(np.argmax(ctd))
            if ctd[i] < tau:
                break
            d_last = ctd[i]
            picks.append(int(lst[i]))
            pending.append(int(lst[i]))
            ctd = np.minimum(ctd, ((cx - cx[i]) ** 2).sum(1).astype(np.float32))
            np_ += 1
        hist_picks.append(np_)
        if adapt:
            if cnt < target[0]:
                eps = min(0.9, eps * 1.3)
            elif cnt > target[1]:
                eps = eps * 0.8
    hp, hl = np.array(hist_picks), np.array(hist_list)
    out = dict(phases=phases, picks_per_phase=(m - 1) / phases, list_mean=hl.mean(), list_p90=np.percentile(hl, 90),
               overflow=overflow, empty=empty, eps_end=eps)
    if bucket:
        out.update(pair_frac=pair_work / pair_all, pairs_per_pick=pair_work / (m - 1),
                   max_wave_pairs_per_phase=float(np.mean(maxwork)))
    return out


if __name__ == "__main__":
    n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    pc = s_scene(3000, 1, n)[0, :, :3].astype(np.float32)
    for cap in (32, 64, 128):
        for eps0 in (0.1, 0.3):
            print("cap", cap, "eps0", eps0, sim(pc, m, eps0, cap, target=(cap // 5, cap * 2 // 3)))
    for bucket in (512, 64):
        for gb in (False, True):
            for il in (False, True):
                print("bucket", bucket, "global_bound", gb, "interleave", il, sim(pc, m, 0.2, 64, bucket=bucket, global_bound=gb, interleave=il))
