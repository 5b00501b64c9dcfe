#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r05_t_pytest_model.txt
