#!/bin/bash
OUT=gpurun_out/r06u; mkdir -p $OUT
for i in 1 2 3; do
timeout 600 python bench.py --train --steps 40 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('train', d['ms_per_step'], d.get('graph_replay_alone_ms'))"
done | tee $OUT/train.txt
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-north-star --no-other-configs --no-other-inputs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('infer', d['ms_per_step'], d['regimes']['throughput_steady_state']['ms_per_step'], d['single_batch_latency_ms'])"
done | tee $OUT/infer.txt
