#!/bin/bash
OUT=gpurun_out/r06ac; mkdir -p $OUT
R=$PWD; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_train -o train -- python $R/bench.py --train --steps 40 --warmup 5 > $R/$OUT/prof_train.log 2>&1); echo "rocprof train rc=$?"
for f in $(find $OUT/prof_train -name "*kernel_stats.csv"); do cp $f $OUT/train_kernel_stats.csv; done
find $OUT -name "*kernel_trace.csv" -delete; rm -rf $OUT/prof_train
