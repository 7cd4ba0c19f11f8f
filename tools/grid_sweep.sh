#!/bin/bash
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-north-star "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['single_batch_latency_ms'])"; }
for rep in 1 2; do for g in 256 248 240 232 224 208 192; do echo "grid=$g $(run --debug-set 6=$g)"; done; done
