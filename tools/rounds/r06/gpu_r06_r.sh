#!/bin/bash
OUT=gpurun_out/r06r; mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for i in 1 2 3; do
timeout 600 python tools/train_flags_ab.py USE_HOIST_BN_STATS=1 -- --steps 40 --warmup 5 2>&1 | tail -1 | tee -a $OUT/ab.txt
timeout 600 python tools/train_flags_ab.py USE_HOIST_BN_STATS=0 -- --steps 40 --warmup 5 2>&1 | tail -1 | tee -a $OUT/ab.txt
done
