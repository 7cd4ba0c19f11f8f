#!/bin/bash
OUT=gpurun_out/r06z; mkdir -p $OUT
PN2_HIP_LIBRARY=$PWD/open3d-pointnet2-semantic3d_amd/libpn2_tune.so timeout 600 python tools/fwd_narrow_ab.py 2>&1 | grep -v amdgpu | tee $OUT/fwd_narrow_ab.txt | tail -14
