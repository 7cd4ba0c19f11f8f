#!/bin/bash
# marginal cost of each entry point in the throughput regime (4 batches in flight): ms/step with the entry launched twice minus
# the baseline; and in the latency regime (one batch in flight)
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-north-star --no-other-configs --no-other-inputs "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r['single_batch_latency_ms'])"; }
echo "base $(run)"
for k in pn2_fps_nested pn2_query_ball_point pn2_three_nn pn2_sa_mlp_max_fused pn2_sa_mlp_fused_pre pn2_sa_mlp_wide_pre pn2_fp_mlp_fused_pre pn2_fp_mlp_wide_pre pn2_fp_interp_concat pn2_linear; do
  echo "$k $(run --dup $k)"
done
echo "base $(run)"
