#!/bin/bash
OUT=gpurun_out/r06ap; mkdir -p $OUT
for v in hip chunk32 chunk16; do echo "== $v"; timeout 300 python tools/wgrad_time.py open3d-pointnet2-semantic3d_amd/libpn2_$v.so 2>&1 | grep -v amdgpu | grep " 8192 x\| 1024 x\| 4096 x\|16384 x\| 32768 x\|total"; done | tee $OUT/chunk.txt
for i in 1 2; do for v in hip chunk32 chunk16; do
PN2_HIP_LIBRARY=$PWD/open3d-pointnet2-semantic3d_amd/libpn2_$v.so timeout 600 python bench.py --train --steps 40 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', d['ms_per_step'], d.get('graph_replay_alone_ms'))"
done; done | tee -a $OUT/chunk.txt
