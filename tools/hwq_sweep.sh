#!/bin/bash
# sweep GPU_MAX_HW_QUEUES x pipeline depth (bench.py throughput), 3 reps each; usage: hwq_sweep.sh
for rep in 1 2 3; do for q in 4 8 16; do for p in 3 4 6 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-north-star --pipeline $p 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('hwq=$q P=$p', r['ms_per_step'], r['single_batch_latency_ms'])"
done; done; done
