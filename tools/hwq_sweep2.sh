#!/bin/bash
# one-stream graphs: pipeline depth sweep at the default HW queue count, fused vs materialised FP front end
for rep in 1 2; do for f in unfused fused; do for p in 1 2 3 4 5 6 8; do
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-north-star --fp-front $f --pipeline $p 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$f P=$p', r['ms_per_step'], r['single_batch_latency_ms'])"
done; done; done
