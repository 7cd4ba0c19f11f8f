#!/bin/bash
OUT=gpurun_out/r06am; mkdir -p $OUT
s=$(date +%s.%N); timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/err.txt; e=$(date +%s.%N); echo "default bench rc=$? seconds $(echo "$e - $s" | bc)"
s=$(date +%s.%N); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/err2.txt; e=$(date +%s.%N); echo "driver-style bench rc=$? seconds $(echo "$e - $s" | bc)"
python -c "
import json
for f in ('bench_default','bench_driver'):
    d=json.loads(open('gpurun_out/r06am/%s.json'%f).read().strip().split('\n')[-1]); print(f, d['value'], d['ms_per_step'], d['steps'], d['warmup'], d['n_gpus'])"
