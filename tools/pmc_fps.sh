#!/bin/bash
mkdir -p gpurun_out/pmc; cd /tmp; export TMPDIR=/tmp
R=$OLDPWD
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc/f1 -o fps -- python $R/tools/fps_ab.py $R/open3d-pointnet2-semantic3d_amd/libpn2_hip.so > $R/gpurun_out/pmc/f1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc/f2 -o fps -- python $R/tools/fps_ab.py $R/open3d-pointnet2-semantic3d_amd/libpn2_hip.so > $R/gpurun_out/pmc/f2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for p in ('f1','f2'):
    for f in glob.glob('gpurun_out/pmc/%s/*counter_collection.csv'%p):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for row in csv.DictReader(open(f)):
            k=row['Kernel_Name'][:70]
            agg[k][row['Counter_Name']]+=float(row['Counter_Value']); n[(k,row['Counter_Name'])]+=1
        for k,v in agg.items():
            if 'fps_reg' in k:
                print(p,k,{c:int(x/n[(k,c)]) for c,x in v.items()})
PY
