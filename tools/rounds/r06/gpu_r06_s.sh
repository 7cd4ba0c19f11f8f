#!/bin/bash
OUT=gpurun_out/r06s; mkdir -p $OUT
timeout 600 python tools/hoist_stats_ab.py 2>&1 | tee $OUT/hoist_stats_ab.txt | tail -6

