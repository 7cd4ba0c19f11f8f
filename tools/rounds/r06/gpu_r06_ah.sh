#!/bin/bash
OUT=gpurun_out/r06ah; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_fullsize_gpu.py tests/test_layers_gpu.py tests/test_model_gpu.py tests/test_runtime_gpu.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4
for i in 1 2 3; do
for v in hip nw4; do
PN2_HIP_LIBRARY=$PWD/open3d-pointnet2-semantic3d_amd/libpn2_$v.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-north-star --no-other-configs --no-other-inputs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
ks={k['kernel']+str(k['args'][:3]):k['avg_us'] for k in d['kernels'] if 'wide' in k['kernel']}
print('$v', d['ms_per_step'], 'steady', d['regimes']['throughput_steady_state']['ms_per_step'], 'lat', d['single_batch_latency_ms'], ks)"
done; done | tee $OUT/ab.txt
