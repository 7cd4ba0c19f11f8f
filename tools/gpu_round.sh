#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench, rocprofv3 kernel trace.  Logs -> gpurun_out/.
# usage: gpurun --timeout 1500 -- 'bash tools/gpu_round.sh <tag> [pytest-args]'
TAG=${1:-r02}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch;print(torch.cuda.get_device_name(0), torch.cuda.device_count())" > $OUT/device.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider "$@" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -40 $OUT/pytest.log
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -3 $OUT/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 3000 $OUT/bench.json; tail -5 $OUT/bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o bench -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OLDPWD/$OUT/prof.log 2>&1); echo "rocprof rc=$?"
find $OUT/prof -name "*stats*" | head; 
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -30 $f; done
# keep the merge-back small: drop the raw per-dispatch trace, keep the stats
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
