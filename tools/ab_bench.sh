#!/bin/bash
# A/B bench helper: runs bench.py several times per variant and prints ms/step.  usage: ab_bench.sh "<flagsA>" "<flagsB>" [reps]
A="$1"; B="$2"; R=${3:-3}
for i in $(seq $R); do
  for v in "$A" "$B"; do
    python bench.py --steps ${STEPS:-40} --warmup 10 --no-cpu-baseline --no-north-star --no-other-configs --no-other-inputs $v 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[%s]' % '$v', r['ms_per_step'], r['single_batch_latency_ms'], (r['regimes'].get('throughput_steady_state') or {}).get('ms_per_step'), r['gpu_ms_per_step_sum_of_kernels'])"
  done
done
