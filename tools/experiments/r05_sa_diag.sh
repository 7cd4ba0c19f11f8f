#!/bin/bash
run() {
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-north-star --no-other-configs --no-other-inputs "$@" 2>/tmp/err.txt \
   | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-80s %.4f  %.4f  %s' % ('$*', r['ms_per_step'], r['single_batch_latency_ms'], (r['regimes'].get('throughput_steady_state') or {}).get('ms_per_step')))" \
   || { echo "FAILED: $*"; tail -5 /tmp/err.txt; }
}
for rep in 1 2; do
run --pipeline 4
run --sampler-ahead 4 --dense-streams 4 --sa-same-stream
run --sampler-ahead 8 --dense-streams 4 --sa-same-stream
run --sampler-ahead 8 --sampler-streams 1 --dense-streams 4
run --sampler-ahead 8 --sampler-streams 1 --dense-streams 2
run --sampler-ahead 8 --sampler-streams 1 --dense-streams 1
run --sampler-ahead 5 --sampler-streams 1 --dense-streams 4
run --sampler-ahead 8 --sampler-streams 3 --dense-streams 3
done
