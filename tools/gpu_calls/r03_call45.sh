#!/bin/bash
# fuzz (240 cases + a second campaign of 480) and the command-line tests with the parallel inflater behind every gzip file
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_cli.py tests/test_gpu_ref_suite.py -q -m gpu -x 2>&1 | tail -15
FLX_FUZZ_CASES=480 FLX_FUZZ_BASE=campaign-2 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -15
