#!/bin/bash
OUT=gpurun_out/r06o; mkdir -p $OUT
PN2_HIP_LIBRARY=$PWD/open3d-pointnet2-semantic3d_amd/libpn2_tune.so PN2_AB_VERBOSE=1 timeout 900 python tools/lin_wres_ab.py > $OUT/lin_wres_ab.txt 2>&1; echo "rc=$?"; tail -60 $OUT/lin_wres_ab.txt
