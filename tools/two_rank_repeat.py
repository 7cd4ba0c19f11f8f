"""Root-cause tool for the two-rank captured-step test: run the world-2 job of
tests/test_train_gpu.py::_ddp_worker N times captured and M times eager, and log for every run which of the test's
assertions would have fired -- replicas bit-identical (captured / eager) versus the trajectory tolerances between two
separately spawned jobs -- plus the pairwise distances between all runs' final parameters (the noise floor the
tolerances have to sit above).

    gpurun --timeout 1800 -- 'python tools/two_rank_repeat.py 30 10 > gpurun_out/two_rank_repeat.txt 2>&1'
"""
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def one_job(capture):
    import torch.multiprocessing as mp
    from test_train_gpu import _ddp_worker
    so = socket.socket(); so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]; so.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, capture, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=400) for _ in range(2))
    for p in procs:
        p.join(60)
    return got


def main():
    n_cap = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    n_eag = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    os.environ["PN2_TWO_RANK_NO_DIAG"] = "1"  # the bench diagnosis legs run after the snapshot: not under test here
    runs = {True: [], False: []}
    for capture, n in ((True, n_cap), (False, n_eag)):
        for i in range(n):
            t0 = time.time()
            got = one_job(capture)
            if any(isinstance(v, str) for v in got.values()):
                print("skip:", got)
                return
            same = bool(np.array_equal(got[0]["p"], got[1]["p"]))
            nd = int((got[0]["p"] != got[1]["p"]).sum())
            print("capture=%d run %2d: replicas identical=%s (differing params %d) split=%s late=%s scale=%s losses0=%s  %.1fs"
                  % (capture, i, same, nd, got[0]["split"], got[0]["late"], got[0]["scale"],
                     ["%.5f" % v for v in got[0]["losses"]], time.time() - t0), flush=True)
            runs[capture].append(got)
    print("identical replicas: captured %d / %d, eager %d / %d"
          % (sum(np.array_equal(g[0]["p"], g[1]["p"]) for g in runs[True]), len(runs[True]),
             sum(np.array_equal(g[0]["p"], g[1]["p"]) for g in runs[False]), len(runs[False])))

    def rel(a, b):
        return float(np.linalg.norm(a - b) / np.linalg.norm(b))

    def loss_rel(a, b):
        return float(np.max(np.abs(np.array(a) - np.array(b)) / np.abs(np.array(b))))
    for name, xs, ys in (("cap-cap", runs[True], runs[True]), ("eag-eag", runs[False], runs[False]),
                         ("cap-eag", runs[True], runs[False])):
        rp, rl, rl2 = [], [], []
        for i, a in enumerate(xs):
            for j, b in enumerate(ys):
                if xs is ys and j <= i:
                    continue
                rp.append(rel(a[0]["p"], b[0]["p"]))
                rl.append(loss_rel(a[0]["losses"], b[0]["losses"]))
                rl2.append(loss_rel(a[0]["losses"][:2], b[0]["losses"][:2]))
        if rp:
            q = lambda v: "min %.3g median %.3g p90 %.3g max %.3g" % (np.min(v), np.median(v), np.percentile(v, 90), np.max(v))  # noqa: E731
            print("%s (%d pairs): param rel [%s]  loss rel (7 steps) [%s]  loss rel (2 eager steps) [%s]" % (name, len(rp), q(rp), q(rl), q(rl2)))
            print("   would fail: rel>5e-2: %d, losses rtol 3e-2: %d, first-two rtol 1e-4: %d"
                  % (sum(v > 5e-2 for v in rp), sum(v > 3e-2 for v in rl), sum(v > 1e-4 for v in rl2)))


if __name__ == "__main__":
    main()
