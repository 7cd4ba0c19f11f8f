#!/bin/bash
OUT=gpurun_out/r06ad; mkdir -p $OUT
for v in hip wide128; do PN2_HIP_LIBRARY=$PWD/open3d-pointnet2-semantic3d_amd/libpn2_$v.so timeout 300 python tools/stats_epilogue_probe.py 2>&1 | tail -1; done | tee $OUT/wide128.txt
for v in hip wide128; do
PN2_HIP_LIBRARY=$PWD/open3d-pointnet2-semantic3d_amd/libpn2_$v.so timeout 600 python bench.py --train --steps 40 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('train', d['ms_per_step'], d.get('graph_replay_alone_ms'))"
done | tee -a $OUT/wide128.txt
