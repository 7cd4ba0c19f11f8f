import importlib.util, os, sys
spec = importlib.util.spec_from_file_location("_b", "/root/repo/open3d-pointnet2-semantic3d_amd/build.py")
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
m.build(force=True)
m.build(extra_flags=["-DPN2_TUNING_HOOKS"], out="/root/repo/open3d-pointnet2-semantic3d_amd/libpn2_tune.so")
print("built")
