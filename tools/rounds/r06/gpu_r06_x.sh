#!/bin/bash
OUT=gpurun_out/r06x; mkdir -p $OUT
echo "== shipped"; timeout 300 python tools/wgrad_time.py 2>&1 | grep -v amdgpu | tee $OUT/wgrad_shipped.txt | tail -26
echo "== plain stores instead of atomics (wrong results: timing only)"; timeout 300 python tools/wgrad_time.py open3d-pointnet2-semantic3d_amd/libpn2_wnoatomic.so 2>&1 | grep -v amdgpu | tee $OUT/wgrad_noatomic.txt | tail -26
