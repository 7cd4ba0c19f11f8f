#!/bin/bash
mkdir -p gpurun_out/pmc; cd /tmp; export TMPDIR=/tmp
R=$OLDPWD
CMD="python $R/bench.py --steps 3 --warmup 1 --eager --no-cpu-baseline --no-north-star --no-other-configs"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc/m1 -o b -- $CMD > $R/gpurun_out/pmc/m1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc/m2 -o b -- $CMD > $R/gpurun_out/pmc/m2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc/m3 -o b -- $CMD > $R/gpurun_out/pmc/m3.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc/m4 -o b -- $CMD > $R/gpurun_out/pmc/m4.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, json
res={}
for p in ('m1','m2','m3','m4'):
    for f in glob.glob('gpurun_out/pmc/%s/*counter_collection.csv'%p):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for row in csv.DictReader(open(f)):
            k=row['Kernel_Name'].replace('(anonymous namespace)::','')[:80]+' grid=%s'%row.get('Grid_Size','')
            agg[k][row['Counter_Name']]+=float(row['Counter_Value']); n[(k,row['Counter_Name'])]+=1
        for k,v in agg.items():
            if 'at::' in k or 'rocclr' in k: continue
            res.setdefault(k,{}).update({c:int(x/n[(k,c)]) for c,x in v.items()})
json.dump(res, open('gpurun_out/pmc/mfma_summary.json','w'), indent=1)
for k,v in res.items():
    if 'linear' in k or 'sa_fused' in k or 'group_point' in k or 'fp_interp' in k:
        wc=v.get('SQ_WAVE_CYCLES',1)
        print(k[:100]); print('   waves',v.get('SQ_WAVES'),'wait_any %.2f wait_inst %.2f active %.2f'%(v.get('SQ_WAIT_ANY',0)/wc, v.get('SQ_WAIT_INST_ANY',0)/wc, v.get('SQ_ACTIVE_INST_ANY',0)/wc), 'mfma_busy',v.get('SQ_VALU_MFMA_BUSY_CYCLES'),'busy',v.get('SQ_BUSY_CYCLES'),'lds_conf',v.get('SQ_LDS_BANK_CONFLICT'),'lds_act',v.get('SQ_ACTIVE_INST_LDS'),'fetchKB',v.get('FETCH_SIZE'),'writeKB',v.get('WRITE_SIZE'))
PY
