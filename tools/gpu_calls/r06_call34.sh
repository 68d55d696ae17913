#!/bin/bash
# round 6, call 34: a longer differential fuzz campaign against the reference binary on the final code (other seeds; stderr compared RAW):
# 960 random invocations (a third of them in k-mer mode: the coverage kernel of round 6, cover_queue.hip, behind the command line),
# 60 damaged gzip cases x five ingest paths, the forked-ranks fuzz
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
FLX_FUZZ_CASES=960 FLX_FUZZ_BASE=r6-campaign FLX_FUZZ_DAMAGED=60 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -12 | tee gpurun_out/r06_call34.log
