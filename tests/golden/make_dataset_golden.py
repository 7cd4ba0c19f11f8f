"""Pins oracle/dataset_oracle.py to the REFERENCE's own code and freezes the result (run in the build container, where
/root/reference exists):

    python tests/golden/make_dataset_golden.py      -> tests/golden/dataset_sampler.npz

dataset/semantic_dataset.py cannot be imported (it imports open3d at module level), so the bodies of
SemanticFileData._get_fix_sized_sample_mask / _center_box / _extract_z_box / sample are lifted out of the reference's
source file with `ast` -- unmodified, never copied into this repository -- and executed on a stand-in object that holds
the same x-sorted arrays.  The restatement must reproduce them bit for bit under the same np.random stream.
"""
import ast
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/dataset/semantic_dataset.py"


def lifted_class():
    tree = ast.parse(open(REF).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "SemanticFileData")
    keep = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in
            ("_get_fix_sized_sample_mask", "_center_box", "_extract_z_box", "sample")]
    assert len(keep) == 4
    mod = ast.Module(body=[ast.ClassDef(name="RefFileData", bases=[], keywords=[], body=keep, decorator_list=[])],
                     type_ignores=[])
    ns = {"np": np}
    exec(compile(ast.fix_missing_locations(mod), REF, "exec"), ns)
    return ns["RefFileData"]


def scene(seed, n):
    rs = np.random.RandomState(seed)
    pts = np.stack([rs.uniform(0, 40, n), rs.uniform(0, 25, n), np.abs(rs.normal(0, 2.0, n))], 1)
    pts = pts.astype(np.float32).astype(np.float64)  # what a float32 .pcd gives Open3D
    labels = rs.randint(0, 9, n).astype(np.int32)
    colors = (rs.randint(0, 256, (n, 3)) / 255.0)
    return pts, labels, colors


def main():
    from oracle.dataset_oracle import FileDataOracle
    Ref = lifted_class()
    out = {}
    for case, (seed, n, npts, box) in enumerate([(0, 60000, 2048, 10), (1, 5000, 4096, 6), (2, 20000, 1024, 3)]):
        pts, labels, colors = scene(seed, n)
        orc = FileDataOracle(pts, labels, colors, box, box)
        ref = Ref()
        ref.box_size_x = ref.box_size_y = box
        ref.points, ref.labels, ref.colors = orc.points, orc.labels, orc.colors  # the same x-sorted arrays (:84-88)
        for k in range(4):
            np.random.seed(100 * case + k)
            r = ref.sample(npts)
            np.random.seed(100 * case + k)
            draws = {}
            o = orc.sample(npts, draws)
            for a, b in zip(r, o):
                assert a.dtype == b.dtype and np.array_equal(a, b), "restatement differs from the reference"
            tag = "c%d_s%d_" % (case, k)
            out[tag + "centered"], out[tag + "raw"], out[tag + "labels"], out[tag + "colors"] = r
            out[tag + "center"] = draws["center"]
            out[tag + "count"] = draws["count"]
            out[tag + "mask"] = draws["mask"] if draws["mask"] is not None else np.zeros(0, dtype=bool)
        out["c%d_meta" % case] = np.array([seed, n, npts, box])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dataset_sampler.npz"), **out)
    print("restatement == lifted reference methods on %d samples; fixture written" % (3 * 4))


if __name__ == "__main__":
    main()
