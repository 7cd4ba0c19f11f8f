#!/bin/bash
# round 6, call B: the on-load batch-norm gradient + glue cuts: tests, then A/B of the training step
OUT=gpurun_out/r06b; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_layers_gpu.py -x -q -m gpu > $OUT/pytest_train_layers.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_train_layers.log
for i in 1 2; do
timeout 600 python tools/train_flags_ab.py USE_BN_GRAD_ON_LOAD=1 2>&1 | tail -1 | tee -a $OUT/ab.txt
timeout 600 python tools/train_flags_ab.py USE_BN_GRAD_ON_LOAD=0 2>&1 | tail -1 | tee -a $OUT/ab.txt
done
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_train -o train -- python $OLDPWD/bench.py --train --steps 20 --warmup 5 > $OLDPWD/$OUT/prof_train.log 2>&1); echo "rocprof train rc=$?"
for f in $(find $OUT/prof_train -name "*kernel_stats.csv"); do cp $f $OUT/train_kernel_stats.csv; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete 2>/dev/null; rm -rf $OUT/prof_train
