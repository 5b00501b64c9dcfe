#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -12 | tee gpurun_out/r05_k_pytest_dist.txt
