#!/bin/bash
# round 6, call E: tests after the finisher / small-K dgrad; training bench with the graph-alone diagnostic; profile
OUT=gpurun_out/r06e; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_layers_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for i in 1 2; do
timeout 600 python bench.py --train --steps 40 --warmup 5 > $OUT/bench_train_$i.json 2> $OUT/bench_train_$i.err
python -c "
import json;d=json.loads(open('$OUT/bench_train_$i.json').read().strip().split('\n')[-1]);print('train ms/step',d['ms_per_step'],'graph alone',d.get('graph_replay_alone_ms'))"
done
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_train -o train -- python $OLDPWD/bench.py --train --steps 95 --warmup 5 > $OLDPWD/$OUT/prof_train.log 2>&1); echo "rocprof train rc=$?"
for f in $(find $OUT/prof_train -name "*kernel_stats.csv"); do cp $f $OUT/train_kernel_stats.csv; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete 2>/dev/null; rm -rf $OUT/prof_train
