#!/bin/bash
OUT=gpurun_out/r06an; mkdir -p $OUT
for i in 1 2 3 4 5; do timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_$i.log 2>&1; echo "run $i rc=$? $(grep -E 'passed|failed' $OUT/pytest_$i.log | tail -1)"; done
