#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x -k forked_ranks 2>&1 | tail -30
