#!/bin/bash
# round 6, call D: the finish-in-producer ticket: tests, A/B, profile
OUT=gpurun_out/r06d; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_layers_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for i in 1 2 3; do
timeout 600 python tools/train_flags_ab.py USE_BN_FINISH_IN_PRODUCER=1 -- --steps 40 --warmup 5 2>&1 | tail -1 | tee -a $OUT/ab.txt
timeout 600 python tools/train_flags_ab.py USE_BN_FINISH_IN_PRODUCER=0 -- --steps 40 --warmup 5 2>&1 | tail -1 | tee -a $OUT/ab.txt
done
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-north-star --no-other-inputs > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python -c "
import json;d=json.loads(open('$OUT/bench.json').read().strip().split('\n')[-1]);print('infer ms/step',d['ms_per_step'],d['ms_per_step_regions'],'lat',d['single_batch_latency_ms'],'steady',d['regimes']['throughput_steady_state'],'verified',d['streams_verified_concurrent'])"
