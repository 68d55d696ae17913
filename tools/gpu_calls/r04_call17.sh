#!/bin/bash
# round 4, call 17: text-matching 12-mers skip the prefilter — parity (k-mer suite) + C3 / C4 cover time
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kmer.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r04_call17.log
echo "== C3" | tee -a gpurun_out/r04_call17.log
timeout 300 python tools/bench_kmer.py --reads 1000000 --steps 3 2>&1 | tail -1 | cut -c1-300 | tee -a gpurun_out/r04_call17.log
echo "== C4" | tee -a gpurun_out/r04_call17.log
timeout 600 python tools/bench_kmer.py --reads 1000000 --steps 3 --trim-split --short-reads 2>&1 | tail -1 | cut -c1-300 | tee -a gpurun_out/r04_call17.log
