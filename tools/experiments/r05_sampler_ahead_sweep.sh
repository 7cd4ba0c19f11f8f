#!/bin/bash
# r05: the throughput regime on runtime.SamplerAheadPipeline (two graphs per batch on sampler / dense streams) against one graph per
# batch; K = 20 like the driver.  Columns: ms/step (K=20), one batch, steady state (200 steps).
# usage: gpurun --timeout 1500 -- 'bash tools/sampler_ahead_sweep.sh > gpurun_out/sa_sweep.txt 2>&1'
run() {  # $1 = env assignments, rest = bench flags
  local envs="$1"; shift
  env $envs python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-north-star --no-other-configs --no-other-inputs "$@" 2>/tmp/err.txt \
   | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-70s %.4f  %.4f  %s' % ('$envs $*', r['ms_per_step'], r['single_batch_latency_ms'], (r['regimes'].get('throughput_steady_state') or {}).get('ms_per_step')))" \
   || { echo "FAILED: $envs $*"; tail -5 /tmp/err.txt; }
}
for rep in 1 2; do
run "A=0" --pipeline 4
run "A=0" --pipeline 8
run "A=0" --sampler-ahead 8 --sampler-streams 2 --dense-streams 2
run "A=0" --sampler-ahead 8 --sampler-streams 1 --dense-streams 3
run "A=0" --sampler-ahead 8 --sampler-streams 2 --dense-streams 3
run "A=0" --sampler-ahead 6 --sampler-streams 2 --dense-streams 2
run "A=0" --sampler-ahead 4 --sampler-streams 2 --dense-streams 2
run "A=0" --sampler-ahead 12 --sampler-streams 2 --dense-streams 2
run "A=0" --sampler-ahead 8 --sampler-streams 4 --dense-streams 4
run "GPU_MAX_HW_QUEUES=8" --sampler-ahead 8 --sampler-streams 4 --dense-streams 4
run "GPU_MAX_HW_QUEUES=8" --sampler-ahead 8 --sampler-streams 2 --dense-streams 3
run "GPU_MAX_HW_QUEUES=8" --pipeline 8
done
# CU split (needs the tuning library for the persistent grid size): samplers on 16 / 32 CUs, dense on the rest
T="PN2_HIP_LIBRARY=$PWD/open3d-pointnet2-semantic3d_amd/libpn2_tune.so"
if [ -f open3d-pointnet2-semantic3d_amd/libpn2_tune.so ]; then
for rep in 1 2; do
run "$T" --pipeline 4
run "$T" --sampler-ahead 8 --sampler-streams 2 --dense-streams 2
run "$T" --sampler-ahead 8 --sampler-streams 2 --dense-streams 2 --cu-split 32 --debug-set 6=224
run "$T" --sampler-ahead 8 --sampler-streams 1 --dense-streams 3 --cu-split 16 --debug-set 6=240
run "$T" --sampler-ahead 8 --sampler-streams 2 --dense-streams 2 --cu-split 32 --debug-set 6=256
done
fi
