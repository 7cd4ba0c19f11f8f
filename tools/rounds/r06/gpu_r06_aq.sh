#!/bin/bash
OUT=gpurun_out/r06aq; mkdir -p $OUT
for i in 1 2; do for v in hip st4 st5; do
PN2_HIP_LIBRARY=$PWD/open3d-pointnet2-semantic3d_amd/libpn2_$v.so timeout 600 python bench.py --train --steps 40 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', d['ms_per_step'], d.get('graph_replay_alone_ms'))"
done; done | tee $OUT/stages.txt
