#!/bin/bash
OUT=gpurun_out/r06ak; mkdir -p $OUT
timeout 600 python examples/predict_synthetic.py > $OUT/predict.txt 2>&1; echo "predict rc=$?"; tail -4 $OUT/predict.txt
timeout 600 python examples/train_synthetic.py > $OUT/train.txt 2>&1; echo "train rc=$?"; tail -5 $OUT/train.txt
