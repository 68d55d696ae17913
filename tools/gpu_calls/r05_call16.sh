#!/bin/bash
# round 5, call 16: a longer differential fuzz campaign against the reference binary on the final code (other seeds; stderr compared RAW):
# 720 random invocations, 90 damaged gzip cases x five ingest paths, the forked-ranks fuzz (rank ranges are the default now)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
FLX_FUZZ_CASES=720 FLX_FUZZ_BASE=r5-campaign FLX_FUZZ_DAMAGED=90 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "random_invocations or damaged_gzip_inputs" 2>&1 | tail -12 | tee gpurun_out/r05_call16.log
