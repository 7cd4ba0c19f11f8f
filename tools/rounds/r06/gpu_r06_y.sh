#!/bin/bash
OUT=gpurun_out/r06y; mkdir -p $OUT
timeout 900 python tools/train_gap_probe.py 2>&1 | grep -v amdgpu | tee $OUT/train_gap.txt | tail -12
