#!/bin/bash
OUT=gpurun_out/r06ao; mkdir -p $OUT
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-north-star --no-other-inputs > $OUT/b1.json 2> $OUT/b1.err; echo "torchrun infer rc=$?"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --train --gpus 1 --steps 5 --warmup 2 > $OUT/t1.json 2> $OUT/t1.err; echo "torchrun train rc=$?"
python -c "
import json
for f in ('b1','t1'):
    d=json.loads(open('gpurun_out/r06ao/%s.json'%f).read().strip().split('\n')[-1]); print(f, d['n_gpus'], d['value'], d['ms_per_step'], d.get('rccl_ranks'), d.get('backend'))"
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2
