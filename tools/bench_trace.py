"""debug: bench.py --train with every step printed"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--train"] + sys.argv[1:]
import bench, torch
import pn2_amd as pn2
orig = pn2.train.Trainer.train_step
def ts(self, *a, **k):
    r = orig(self, *a, **k)
    print("step", self.step_count, r, "graph" if self._graph is not None else "eager", flush=True)
    return r
pn2.train.Trainer.train_step = ts
bench.main()
