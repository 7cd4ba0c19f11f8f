#!/bin/bash
# bench at several pipeline depths (one gpurun call)
mkdir -p gpurun_out/sweep
for p in 1 2 3 4; do
  timeout 300 python bench.py --steps 40 --warmup 8 --pipeline $p --no-cpu-baseline --no-north-star 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('pipeline', $p, 'ms/step', r['ms_per_step'], 'Mpts/s', round(r['value']/1e6,1))" | tee -a gpurun_out/sweep/pipeline.txt
done
timeout 300 python bench.py --steps 40 --warmup 8 --pipeline 1 --one-stream --no-cpu-baseline --no-north-star 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('one-stream p1 ms/step', r['ms_per_step'])" | tee -a gpurun_out/sweep/pipeline.txt
