#!/bin/bash
# One gpurun call: smoke, bench (driver flags), rocprofv3 kernel stats of the same command, PMC passes (separate runs, counters
# never combined with a trace domain other than --kernel-trace), PMC of the two north-star kernel shapes.
# usage: gpurun --timeout 2400 -- 'bash tools/gpu_round6.sh r06'
TAG=${1:-r06}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
python -c "import torch;print(torch.cuda.get_device_name(0), torch.cuda.device_count())" > $OUT/device.txt 2>&1
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 600 python bench.py --train --steps 20 --warmup 5 > $OUT/bench_train.json 2> $OUT/bench_train.err; echo "bench train rc=$?"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-north-star --no-other-inputs > $R/$OUT/prof.log 2>&1); echo "rocprof rc=$?"
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_train -o train -- python $R/bench.py --train --steps 20 --warmup 5 > $R/$OUT/prof_train.log 2>&1); echo "rocprof train rc=$?"
for f in $(find $OUT/prof_train -name "*kernel_stats.csv"); do cp $f $OUT/train_kernel_stats.csv; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete 2>/dev/null
rm -rf gpurun_out/pmc; timeout 900 bash tools/pmc_mfma.sh > $OUT/pmc.log 2>&1; echo "pmc rc=$?"
cp gpurun_out/pmc/mfma_summary.json $OUT/pmc_summary.json 2>/dev/null
rm -rf gpurun_out/pmc
# north-star kernel shapes (group_point at C=128, the 131 -> 128 fused layer + max, ball query): same four PMC passes
mkdir -p gpurun_out/pmc
NS="python $R/bench.py --only-north-star"
(cd /tmp
 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc/n1 -o b -- $NS > $R/gpurun_out/pmc/n1.log 2>&1
 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc/n3 -o b -- $NS > $R/gpurun_out/pmc/n3.log 2>&1
 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc/n4 -o b -- $NS > $R/gpurun_out/pmc/n4.log 2>&1
 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmc/n5 -o b -- $NS > $R/gpurun_out/pmc/n5.log 2>&1)
python - <<'PY'
import csv, glob, collections, json
res = {}
for p in ('n1', 'n3', 'n4'):
    for f in glob.glob('gpurun_out/pmc/%s/*counter_collection.csv' % p):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name'].replace('(anonymous namespace)::', '')[:90] + ' grid=%s' % row.get('Grid_Size', '')
            agg[k][row['Counter_Name']] += float(row['Counter_Value']); n[(k, row['Counter_Name'])] += 1
        for k, v in agg.items():
            if 'at::' in k or 'rocclr' in k: continue
            res.setdefault(k, {}).update({c: int(x / n[(k, c)]) for c, x in v.items()})
for f in glob.glob('gpurun_out/pmc/n5/*kernel_stats.csv'):
    for row in csv.DictReader(open(f)):
        nm = row['Name'].replace('(anonymous namespace)::', '')[:90]
        for k in res:
            if k.startswith(nm[:60]): res[k]['avg_ns_rocprofv3'] = float(row['AverageNs']); res[k]['calls'] = int(row['Calls'])
json.dump(res, open('gpurun_out/pmc/north_star_summary.json', 'w'), indent=1)
PY
cp gpurun_out/pmc/north_star_summary.json $OUT/pmc_north_star.json 2>/dev/null
rm -rf gpurun_out/pmc/n1 gpurun_out/pmc/n3 gpurun_out/pmc/n4 gpurun_out/pmc/n5
ls -la $OUT
