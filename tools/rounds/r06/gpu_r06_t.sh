#!/bin/bash
OUT=gpurun_out/r06t; mkdir -p $OUT
for v in hip f32 nomath noatomic; do PN2_HIP_LIBRARY=$PWD/open3d-pointnet2-semantic3d_amd/libpn2_$v.so timeout 300 python tools/stats_epilogue_probe.py 2>&1 | tail -1; done | tee $OUT/stats_epilogue.txt
