#!/bin/bash
# One gpurun call: smoke, bench (driver flags), rocprofv3 kernel stats of the same command, PMC passes (separate runs).
# usage: gpurun --timeout 1500 -- 'bash tools/gpu_round3.sh <tag>'
TAG=${1:-r03}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
python -c "import torch;print(torch.cuda.get_device_name(0), torch.cuda.device_count())" > $OUT/device.txt 2>&1
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-north-star > $R/$OUT/prof.log 2>&1); echo "rocprof rc=$?"
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_train -o train -- python $R/bench.py --train --steps 20 --warmup 5 > $R/$OUT/prof_train.log 2>&1); echo "rocprof train rc=$?"
for f in $(find $OUT/prof_train -name "*kernel_stats.csv"); do cp $f $OUT/train_kernel_stats.csv; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete 2>/dev/null
rm -rf gpurun_out/pmc; timeout 900 bash tools/pmc_mfma.sh > $OUT/pmc.log 2>&1; echo "pmc rc=$?"
cp gpurun_out/pmc/mfma_summary.json $OUT/pmc_summary.json 2>/dev/null
rm -rf gpurun_out/pmc/m1 gpurun_out/pmc/m2 gpurun_out/pmc/m3 gpurun_out/pmc/m4
ls -la $OUT
