#!/bin/bash
# round 6, call A: today's numbers on this box + the glue probe + the two-rank repeat
OUT=gpurun_out/r06a; mkdir -p $OUT
python -c "import torch;print(torch.cuda.get_device_name(0), torch.cuda.device_count())" > $OUT/device.txt 2>&1
timeout 600 python bench.py --train --steps 20 --warmup 5 > $OUT/bench_train.json 2> $OUT/bench_train.err; echo "bench train rc=$?"
timeout 600 python tools/train_glue_probe.py > $OUT/glue.txt 2>&1; echo "glue rc=$?"
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 1500 python tools/two_rank_repeat.py 30 10 > $OUT/two_rank_repeat.txt 2>&1; echo "two-rank rc=$?"
tail -12 $OUT/two_rank_repeat.txt
