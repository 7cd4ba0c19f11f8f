#!/bin/bash
OUT=gpurun_out/r06g; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_layers_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
for i in 1 2 3; do
timeout 600 python bench.py --train --steps 40 --warmup 5 > $OUT/bench_train_$i.json 2> $OUT/bench_train_$i.err
python -c "
import json;d=json.loads(open('$OUT/bench_train_$i.json').read().strip().split('\n')[-1]);print('train ms/step',d['ms_per_step'],'graph alone',d.get('graph_replay_alone_ms'))"
done
