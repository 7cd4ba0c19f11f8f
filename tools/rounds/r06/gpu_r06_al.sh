#!/bin/bash
OUT=gpurun_out/r06al; mkdir -p $OUT
timeout 900 python -m pytest tests/test_layers_gpu.py tests/test_train_gpu.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
for i in 1 2 3; do timeout 600 python bench.py --train --steps 40 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('train', d['ms_per_step'], d.get('graph_replay_alone_ms'))"; done | tee $OUT/train.txt
