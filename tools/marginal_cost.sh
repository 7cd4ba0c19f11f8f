#!/bin/bash
# marginal cost of each entry point in the pipelined regime: ms/step with the entry launched twice minus baseline
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-north-star "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"; }
for rep in 1 2; do
echo "base $(run)"
for k in pn2_fps_gather pn2_query_ball_point pn2_sa_mlp_max_fused pn2_sa_mlp_rows_fused pn2_linear pn2_three_nn pn2_fp_interp_concat pn2_fp_mlp_fused pn2_mlp_chain pn2_sa_group_concat; do
  echo "$k $(run --dup $k)"
done; done
