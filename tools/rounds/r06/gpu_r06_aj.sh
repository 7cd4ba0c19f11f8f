#!/bin/bash
for i in 1 2 3; do timeout 600 python bench.py --train --steps 40 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('train', d['ms_per_step'], d.get('graph_replay_alone_ms'))"; done
