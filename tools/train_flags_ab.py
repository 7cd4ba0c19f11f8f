"""A/B of the training-path switches: python tools/train_flags_ab.py NAME=0|1 ... [-- bench args]
e.g. python tools/train_flags_ab.py USE_BN_ON_LOAD=1 -- --steps 20 --warmup 5"""
import json, os, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pn2_amd as pn2
args = sys.argv[1:]
rest = args[args.index("--") + 1:] if "--" in args else ["--steps", "20", "--warmup", "5"]
flags = [a for a in (args[:args.index("--")] if "--" in args else args) if "=" in a]
for f in flags:
    k, v = f.split("=")
    for mod in (pn2.util.tf_util, pn2.util.pointnet_util):
        if hasattr(mod, k):
            setattr(mod, k, bool(int(v)))
import bench
sys.argv = ["bench.py", "--train"] + rest
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(" ".join(flags) or "defaults", "->", d["ms_per_step"], "ms per step")
