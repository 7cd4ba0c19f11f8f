"""Importable alias of the package directory `open3d-pointnet2-semantic3d_amd/`
(its mandated name is not a Python identifier):

    import pn2_amd as pn2
    pn2.farthest_point_sample(...), pn2.util.pointnet_util.pointnet_sa_module(...)

`pn2_amd` IS that package object (one copy of every submodule); reach submodules
by attribute access, e.g. pn2.tf_ops.tf_sampling.
"""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
_pkg = importlib.import_module("open3d-pointnet2-semantic3d_amd")
sys.modules[__name__] = _pkg
