#!/bin/bash
OUT=gpurun_out/r06q; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_bench_contract_gpu.py tests/test_train_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -2 $OUT/bench.err
timeout 600 python bench.py --train --steps 20 --warmup 5 > $OUT/bench_train.json 2> $OUT/bench_train.err; echo "train rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06q/bench.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['ms_per_step_regions'], d['single_batch_latency_ms'], d['roofline']['traffic'], d['roofline']['traffic_source'])
t=json.loads(open('gpurun_out/r06q/bench_train.json').read().strip().split('\n')[-1]); print(t['ms_per_step'])
PY
