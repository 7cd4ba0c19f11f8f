#!/bin/bash
# timeline of a few training steps (kernel trace) -> gpurun_out/r06i/timeline.txt.gz
OUT=gpurun_out/r06i; mkdir -p $OUT; R=$PWD; export TMPDIR=/tmp
rm -rf /tmp/tr_i
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_i -o t -- python $R/bench.py --train --steps 10 --warmup 5 > $R/$OUT/bench.json 2> /tmp/tr_i.err)
f=$(find /tmp/tr_i -name "*kernel_trace.csv" | head -1)
python tools/trace_timeline.py $f --last-ms 60 > $OUT/timeline.txt
gzip -f $OUT/timeline.txt
ls -la $OUT
