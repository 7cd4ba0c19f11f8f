#!/bin/bash
run() {
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-north-star --no-other-configs --no-other-inputs "$@" 2>/tmp/err.txt \
   | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %.4f  %.4f  %s' % ('$*', r['ms_per_step'], r['single_batch_latency_ms'], (r['regimes'].get('throughput_steady_state') or {}).get('ms_per_step')))" \
   || { echo "FAILED: $*"; tail -5 /tmp/err.txt; }
}
for rep in 1 2 3; do
run --pipeline 4
run --stagger 0,0,0,0
run --stagger 0,0,1,1
run --stagger 0,1,1,2
run --stagger 0,1,2,3
run --stagger 0,1,0,1
run --stagger 0,0,0,1
run --stagger 1,1,1,1
run --stagger 0,0,1
run --stagger 0,1,2
done
