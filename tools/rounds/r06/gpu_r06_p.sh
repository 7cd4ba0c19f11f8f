#!/bin/bash
OUT=gpurun_out/r06p; mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
for i in ; do
for f in "" "--no-binned-bq"; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-north-star --no-other-configs --no-other-inputs $f > $OUT/b.json 2>$OUT/b.err
python - "$f" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r06p/b.json').read().strip().split('\n')[-1])
print('%-16s' % (sys.argv[1] or 'binned'), d['ms_per_step'], d['ms_per_step_regions'], 'steady', d['regimes']['throughput_steady_state']['ms_per_step'], 'lat', d['single_batch_latency_ms'])
PY
done; done | tee $OUT/ab.txt
