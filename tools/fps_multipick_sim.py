"""How many exact FPS picks could one synchronised round deliver?  After the winner p1 of a round is known, the runner-up r
(2nd largest running-min distance) IS the next pick whenever dist(r, p1)^2 >= td[r] (its td survives the update unchanged and
every other td can only fall) -- and so on down the ranking.  Numpy simulation on the bench's S-scene (statistics only: ties are
broken by index order here, not by the reference's (k mod 512, k) rule).  Result quoted in DESIGN.md section 9."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import s_scene


def sim(x, m, kmax):
    n = len(x)
    td = np.full(n, 1e38, np.float32)
    npicks, cur, rounds, hist = 1, 0, 0, np.zeros(kmax + 1, int)
    td = np.minimum(td, ((x - x[cur]) ** 2).sum(1).astype(np.float32))
    while npicks < m:
        order = np.argsort(-td, kind="stable")[:kmax]
        chain = [order[0]]
        for r in order[1:]:
            if all(((x[r] - x[p]) ** 2).sum() >= td[r] for p in chain):
                chain.append(r)
            else:
                break
        chain = chain[:m - npicks]
        for p in chain:
            td = np.minimum(td, ((x - x[p]) ** 2).sum(1).astype(np.float32))
        npicks += len(chain)
        hist[len(chain)] += 1
        rounds += 1
    return rounds, hist


pc = s_scene(3000, 1, 8192)[0, :, :3].astype(np.float32)
for kmax in (1, 2, 3, 4):
    r, h = sim(pc, 1024, kmax)
    print("n=8192 m=1024, up to %d picks per round: %d rounds for 1023 picks (%.2f per round), histogram %s" % (kmax, r, 1023 / r, h[1:].tolist()))
