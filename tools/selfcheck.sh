#!/bin/bash
# Which of the tools/ scripts still start and run against the current library?  Every python tool gets 25 s (the tuning library for
# those that use pn2_debug_set hooks): rc 0 = finished, 124 = still running when the time was up (it runs), else broken.
# usage: gpurun --timeout 1500 -- 'bash tools/selfcheck.sh > gpurun_out/tools_selfcheck.txt 2>&1'
T=$PWD/open3d-pointnet2-semantic3d_amd/libpn2_tune.so
for f in tools/*.py; do
  b=$(basename $f)
  case $b in __init__.py|pmc_to_profiles.py|trace_timeline.py) continue;; esac
  if grep -q "pn2_debug_set\|debug_set_" $f; then lib="PN2_HIP_LIBRARY=$T"; tag=tuning-lib; else lib="X=0"; tag=shipped-lib; fi
  s=$(date +%s)
  env $lib timeout 25 python $f > /tmp/sc.out 2>&1; rc=$?
  e=$(date +%s)
  printf "%-28s rc %3d  %3d s  %-11s  %s\n" $b $rc $((e - s)) $tag "$(grep -v amdgpu.ids /tmp/sc.out | tail -1 | cut -c1-110)"
done
