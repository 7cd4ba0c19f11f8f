for w in auto off 4096; do for P in 1 4 8; do
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-north-star --wide $w --pipeline $P 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wide=$w P=$P', d['ms_per_step'], d['regimes']['latency']['ms_per_step'])"
done; done
