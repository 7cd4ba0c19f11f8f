#!/bin/bash
# PMC passes over a short training run (separate rocprofv3 runs per counter group, --kernel-trace only beside them) ->
# gpurun_out/pmc_train/summary.json: per kernel the averaged counters, MFMA busy fraction, HBM bytes per launch (gfx950: FETCH_SIZE
# x 2, as tools/pmc_to_profiles.py).   usage: gpurun --timeout 1500 -- 'bash tools/pmc_train.sh'
mkdir -p gpurun_out/pmc_train; R=$PWD; cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --train --steps 6 --warmup 2"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_train/t1 -o b -- $CMD > $R/gpurun_out/pmc_train/t1.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_train/t2 -o b -- $CMD > $R/gpurun_out/pmc_train/t2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_train/t3 -o b -- $CMD > $R/gpurun_out/pmc_train/t3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmc_train/t4 -o b -- $CMD > $R/gpurun_out/pmc_train/t4.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, json
res = {}
for p in ('t1', 't2', 't3'):
    for f in glob.glob('gpurun_out/pmc_train/%s/*counter_collection.csv' % p):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name'].replace('(anonymous namespace)::', '')[:70] + ' grid=%s' % row.get('Grid_Size', '')
            agg[k][row['Counter_Name']] += float(row['Counter_Value']); n[(k, row['Counter_Name'])] += 1
        for k, v in agg.items():
            if 'at::' in k or 'rocclr' in k: continue
            res.setdefault(k, {}).update({c: int(x / n[(k, c)]) for c, x in v.items()})
for f in glob.glob('gpurun_out/pmc_train/t4/*kernel_stats.csv'):
    for row in csv.DictReader(open(f)):
        nm = row['Name'].replace('(anonymous namespace)::', '')[:70]
        for k in res:
            if k.startswith(nm[:60]): res[k]['avg_ns_rocprofv3'] = float(row['AverageNs']); res[k]['calls'] = int(row['Calls'])
for k, v in res.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and v.get('GRBM_GUI_ACTIVE'):
        v['mfma_busy_frac'] = round(v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * v['GRBM_GUI_ACTIVE'] / 8.0), 4)
    if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
        v['hbm_bytes'] = 2 * v['FETCH_SIZE'] * 1024 + v['WRITE_SIZE'] * 1024
json.dump(res, open('gpurun_out/pmc_train/summary.json', 'w'), indent=1)
top = sorted(res.items(), key=lambda kv: -kv[1].get('avg_ns_rocprofv3', 0) * kv[1].get('calls', 0))[:14]
for k, v in top: print('%-80s avg %.1f us  mfma_busy %s  hbm %s MB' % (k[:80], v.get('avg_ns_rocprofv3', 0) / 1e3, v.get('mfma_busy_frac'), None if 'hbm_bytes' not in v else round(v['hbm_bytes'] / 1e6, 1)))
PY
rm -rf gpurun_out/pmc_train/t1 gpurun_out/pmc_train/t2 gpurun_out/pmc_train/t3 gpurun_out/pmc_train/t4
