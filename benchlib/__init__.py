"""Pieces of bench.py (the driver contract, the timed regions and the CPU baseline stay in bench.py at the repository root):
inputs -- the synthetic clouds; accounting -- algorithmic work per launch, roofline constants, the HIP-event trace summary;
configs -- the kernel-level north-star targets and BASELINE configs[2] / configs[4] in both regimes; training -- the `--train` line
(BASELINE configs[3]) and its multi-rank decomposition; launch -- self-launch under torch.distributed.run, `--dry-run`."""
