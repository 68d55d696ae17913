#!/bin/bash
# round 4, call 1: the parity items of the round-3 review — every register-history instantiation, damaged gzip, --verbose with ranks
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_phred.py tests/test_gpu_fuzz.py tests/test_gpu_cli.py -x -q -m gpu --durations=15 2>&1 | tail -40 | tee gpurun_out/r04_call1.log
