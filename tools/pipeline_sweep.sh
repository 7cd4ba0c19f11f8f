#!/bin/bash
# bench at several pipeline depths (one gpurun call): the driver's K = 20 / W = 5 and a long run; 2 repetitions each
mkdir -p gpurun_out/sweep
: > gpurun_out/sweep/pipeline.txt
for rep in 1 2; do
for p in ${PIPES:-3 4 5 6 8}; do
  for kw in "20 5" "200 20"; do
    set -- $kw
    timeout 300 python bench.py --steps $1 --warmup $2 --pipeline $p --no-cpu-baseline --no-north-star --no-other-configs --no-other-inputs 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipeline', $p, 'K', $1, 'ms/step', r['ms_per_step'], 'Mpts/s', round(r['value']/1e6,1))" | tee -a gpurun_out/sweep/pipeline.txt
  done
done
done
