#!/bin/bash
# round 4, call 18: a longer differential fuzz campaign against the reference binary (other seeds): 960 random invocations, 180 damaged gzip cases
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
FLX_FUZZ_CASES=960 FLX_FUZZ_BASE=r4-campaign FLX_FUZZ_DAMAGED=180 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "random_invocations_match or damaged_gzip_inputs_match" 2>&1 | tail -30 | tee gpurun_out/r04_call18.log
