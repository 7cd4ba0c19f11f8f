#!/bin/bash
OUT=gpurun_out/r06k; mkdir -p $OUT
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06k/bench.json').read().strip().split('\n')[-1])
print('ms/step',d['ms_per_step'],d['ms_per_step_regions'],'lat',d['single_batch_latency_ms'])
oc=d.get('other_configs',{})
print(json.dumps({k:(v.get('regimes') if isinstance(v,dict) else v) for k,v in oc.items()},indent=1)[:1500])
print('cfg4 B16', json.dumps(oc.get('configs[4]',{}).get('B16'),indent=1)[:1200])
print('cfg3', oc.get('configs[3]@1gpu'))
print('err', oc.get('error'))
PY
tail -3 $OUT/bench.err
