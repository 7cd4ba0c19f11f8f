#!/bin/bash
# usage: resusage.sh file.hip [filter]
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-gpu-rdc -fno-slp-vectorize -Wno-unused-function -c "$1" -o /tmp/resusage.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur=None;rows=[]
for l in sys.stdin:
    if 'error' in l: print(l.strip())
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'n':m.group(1)};rows.append(cur)
    for k,pat in (('v','    VGPRs: (\d+)'),('a','AGPRs: (\d+)'),('sp','VGPRs Spill: (\d+)'),('occ','Occupancy \[waves/SIMD\]: (\d+)'),('lds','LDS Size \[bytes/block\]: (\d+)')):
        m=re.search(pat,l)
        if m and cur is not None: cur[k]=m.group(1)
names=subprocess.run(['/usr/bin/c++filt']+[r['n'] for r in rows],capture_output=True,text=True).stdout.split('\n')
flt=sys.argv[1] if len(sys.argv)>1 else ''
for r,n in zip(rows,names):
    n=n.replace('(anonymous namespace)::','')
    n=re.sub(r'\(.*','',n)
    if flt in n: print('%-70s v=%s a=%s spill=%s occ=%s lds=%s'%(n[:70],r.get('v'),r.get('a'),r.get('sp'),r.get('occ'),r.get('lds')))
" "$2"
