#!/bin/bash
# round 6, call C: pooled reduce + widths + fps m>n tests; A/B
OUT=gpurun_out/r06c; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_layers_gpu.py tests/test_ref_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest.log
for i in 1 2; do
timeout 600 python tools/train_flags_ab.py USE_POOLED_BN_REDUCE=1 -- --steps 40 --warmup 5 2>&1 | tail -1 | tee -a $OUT/ab.txt
timeout 600 python tools/train_flags_ab.py USE_POOLED_BN_REDUCE=0 -- --steps 40 --warmup 5 2>&1 | tail -1 | tee -a $OUT/ab.txt
done
