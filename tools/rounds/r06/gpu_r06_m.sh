#!/bin/bash
OUT=gpurun_out/r06m; mkdir -p $OUT
for w in both coarse nn none; do echo "== side: $w"; PN2_SIDE_WHAT=$w timeout 600 python tools/latency_branch_ab.py 2>&1 | tail -3; done | tee $OUT/latency_branch.txt
