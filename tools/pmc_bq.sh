#!/bin/bash
# PMC counters for the ball query / three_nn kernels (separate passes, no tracing domains)
mkdir -p gpurun_out/pmc; cd /tmp; export TMPDIR=/tmp
R=$OLDPWD
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc/p1 -o bq -- python $R/tools/bq_ab.py > $R/gpurun_out/pmc/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $R/gpurun_out/pmc/p2 -o bq -- python $R/tools/bq_ab.py > $R/gpurun_out/pmc/p2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for p in ('p1','p2'):
    for f in glob.glob('gpurun_out/pmc/%s/*counter_collection.csv'%p):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for row in csv.DictReader(open(f)):
            k=row['Kernel_Name'][:60]
            agg[k][row['Counter_Name']]+=float(row['Counter_Value']); 
        for k,v in agg.items():
            if 'ball_query' in k or 'three_nn' in k:
                print(p,k,{c:int(x) for c,x in v.items()})
PY
