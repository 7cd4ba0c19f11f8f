#!/bin/bash
# usage: trace_run.sh <tag> <bench flags...>  -> gpurun_out/<tag>_timeline.txt (last 12 ms of the run's kernel trace)
TAG=$1; shift
R=$PWD; export TMPDIR=/tmp
rm -rf /tmp/tr_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$TAG -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-north-star --no-other-configs --no-other-inputs "$@" > $R/gpurun_out/${TAG}_bench.json 2> /tmp/tr_$TAG.err)
f=$(find /tmp/tr_$TAG -name "*kernel_trace.csv" | head -1)
python tools/trace_timeline.py $f > gpurun_out/${TAG}_timeline_full.txt
gzip -f gpurun_out/${TAG}_timeline_full.txt
