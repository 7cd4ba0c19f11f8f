#!/bin/bash
timeout 900 python -m pytest tests/test_layers_gpu.py -x -q -m gpu -k "wgrad or eight_wave" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12
