#!/bin/bash
# usage: tools/dbg/isa.sh <file.hip> <mangled-kernel-substring> [out.s]  -- compile one csrc file, extract one kernel's ISA
cd /root/repo/open3d-pointnet2-semantic3d_amd/csrc || exit 1
F=$1; K=$2; OUT=${3:-/tmp/kernel.s}
mkdir -p /tmp/isa && cd /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-gpu-rdc -fno-slp-vectorize -Wall -Wno-unused-function $EXTRA -c /root/repo/open3d-pointnet2-semantic3d_amd/csrc/$F -o /tmp/isa/out.o -save-temps=obj 2>&1 | grep -E "error|warning: [^u]"
S=$(ls -t /tmp/isa/*gfx950.s | head -1)
a=$(grep -n "^_Z[^ ]*${K}[^ ]*:" $S | head -1 | cut -d: -f1)
b=$(grep -n "amdhsa_kernel _Z.*${K}" $S | head -1 | cut -d: -f1)
[ -z "$a" ] && { echo "kernel not found"; exit 1; }
sed -n ${a},${b}p $S > $OUT
grep -E "\.num_vgpr|\.private_seg_size|numbered_sgpr" $S | grep "$K" | head -3
wc -l $OUT
