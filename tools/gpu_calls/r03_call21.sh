#!/bin/bash
# differential fuzz: 240 seeded random invocations, new binary vs the reference binary
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -60 > gpurun_out/c21_fuzz.log
cat gpurun_out/c21_fuzz.log
