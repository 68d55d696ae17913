#!/bin/bash
# round 4, call 22: one output file written by all forked ranks — the sinks test, the multi-rank CLI tests, the ranks fuzz
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_comm2.py tests/test_gpu_fuzz.py -x -q -m gpu -k "sinks or two_ranks or forked_ranks or launch_fails or verbose" 2>&1 | tail -12 | tee gpurun_out/r04_call22.log
