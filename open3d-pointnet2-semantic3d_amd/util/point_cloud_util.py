"""Label / point-cloud file helpers of the reference (util/point_cloud_util.py:53-63) plus a minimal PCD reader/writer
(the reference goes through Open3D's read_point_cloud / write_point_cloud, which is not available here)."""
import struct

import numpy as np


def load_labels(label_path):
    """one int per line -> int32 array (point_cloud_util.py:53-57)"""
    with open(label_path, "r") as f:
        return np.array([int(line) for line in f], dtype=np.int32)


def write_labels(label_path, labels):
    """point_cloud_util.py:60-63"""
    with open(label_path, "w") as f:
        f.write("".join("%d\n" % l for l in np.asarray(labels).tolist()))


def write_point_cloud_pcd(path, points, colors=None, binary=True):
    """PCD v0.7 with FIELDS x y z [rgb] (float32; rgb packed 0x00RRGGBB in a float, as Open3D / PCL write it)."""
    points = np.asarray(points, dtype=np.float32)
    n = len(points)
    has_c = colors is not None
    fields = "x y z rgb" if has_c else "x y z"
    header = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS %s\nSIZE %s\nTYPE %s\nCOUNT %s\n"
              "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n" % (
                  fields, " ".join(["4"] * (4 if has_c else 3)), " ".join(["F"] * (4 if has_c else 3)),
                  " ".join(["1"] * (4 if has_c else 3)), n, n, "binary" if binary else "ascii"))
    cols = [points]
    if has_c:
        c8 = np.clip(np.floor(np.asarray(colors, dtype=np.float64) * 255.0), 0, 255).astype(np.uint32)
        packed = ((c8[:, 0] << 16) | (c8[:, 1] << 8) | c8[:, 2]).astype(np.uint32).view(np.float32)
        cols.append(packed[:, None])
    data = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype=np.float32)
    with open(path, "wb") as f:
        f.write(header.encode())
        if binary:
            f.write(data.tobytes())
        else:
            for row in data:
                f.write((" ".join(repr(float(v)) for v in row) + "\n").encode())


def read_point_cloud_pcd(path):
    """-> points (n,3) float64, colors (n,3) float64 in [0,1] (zeros when the file has no rgb), like np.asarray(pcd.points)
    / np.asarray(pcd.colors) after open3d.read_point_cloud."""
    with open(path, "rb") as f:
        meta = {}
        while True:
            line = f.readline().decode("ascii", "replace").strip()
            if not line or line.startswith("#"):
                if not line:
                    raise ValueError("truncated PCD header")
                continue
            key, _, val = line.partition(" ")
            meta[key] = val.split()
            if key == "DATA":
                break
        fields, n = meta["FIELDS"], int(meta["POINTS"][0])
        sizes = [int(s) for s in meta["SIZE"]]
        if any(s != 4 for s in sizes) or any(t not in "FUI" for t in meta["TYPE"]):
            raise ValueError("only 4-byte PCD fields are supported")
        if meta["DATA"][0] == "binary":
            raw = np.frombuffer(f.read(n * 4 * len(fields)), dtype=np.float32).reshape(n, len(fields))
        elif meta["DATA"][0] == "ascii":
            raw = np.array([[struct.unpack("f", struct.pack("f", float(v)))[0] for v in f.readline().split()]
                            for _ in range(n)], dtype=np.float32).reshape(n, len(fields))
        else:
            raise ValueError("unsupported PCD DATA %s" % meta["DATA"][0])
    xyz = np.stack([raw[:, fields.index(a)] for a in "xyz"], axis=1).astype(np.float64)
    if "rgb" in fields:
        p = np.ascontiguousarray(raw[:, fields.index("rgb")]).view(np.uint32)
        colors = np.stack([(p >> 16) & 255, (p >> 8) & 255, p & 255], axis=1).astype(np.float64) / 255.0
    else:
        colors = np.zeros_like(xyz)
    return xyz, colors
