#!/bin/bash
timeout 900 python -m pytest tests/test_layers_gpu.py -x -q -m gpu -k "streaming_forward or ticket" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8
