#!/bin/bash
OUT=gpurun_out/r06v; mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
