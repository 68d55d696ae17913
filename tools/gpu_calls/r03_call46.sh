#!/bin/bash
# header-only records (stale quality / dead stream), then the fuzz campaigns again
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x -k header_only 2>&1 | tail -25
timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k random 2>&1 | tail -25
FLX_FUZZ_CASES=480 FLX_FUZZ_BASE=campaign-2 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k random 2>&1 | tail -40
