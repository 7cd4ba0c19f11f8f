#!/bin/bash
# kernel timeline of the last training steps (per-launch durations with grid sizes)
OUT=gpurun_out/r06n; mkdir -p $OUT
R=$PWD; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_train -o t -- python $R/bench.py --train --steps 10 --warmup 5 > $R/$OUT/bench.json 2> $R/$OUT/tr.err)
f=$(find /tmp/tr_train -name "*kernel_trace.csv" | head -1)
python tools/trace_timeline.py $f --last-ms 8 > $OUT/train_timeline.txt
wc -l $OUT/train_timeline.txt
