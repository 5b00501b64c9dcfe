#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -s -k "train_eval_parity and b32" 2>&1 | grep "elementwise\|passed\|failed\|worst" | tee gpurun_out/r05_n_fullsize_elementwise.txt
