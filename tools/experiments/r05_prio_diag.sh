#!/bin/bash
run() {
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-north-star --no-other-configs --no-other-inputs "$@" 2>/tmp/err.txt \
   | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-80s %.4f  %.4f  %s' % ('$*', r['ms_per_step'], r['single_batch_latency_ms'], (r['regimes'].get('throughput_steady_state') or {}).get('ms_per_step')))" \
   || { echo "FAILED: $*"; tail -5 /tmp/err.txt; }
}
python -c "
import torch
print('priority range', torch.cuda.Stream.priority_range())"
for rep in 1 2 3; do
run --pipeline 4
run --pipeline 4 --stream-priorities=-1,0,0,1
run --pipeline 4 --stream-priorities=-1,-1,0,0
run --pipeline 4 --stream-priorities=-1,0,0,0
run --pipeline 4 --stream-priorities=1,0,0,0
run --pipeline 3 --stream-priorities=-1,0,1
run --pipeline 5 --stream-priorities=-1,0,0,1,1
done
