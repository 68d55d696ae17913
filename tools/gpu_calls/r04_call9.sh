#!/bin/bash
# round 4, call 9: two-round prefilter — parity (k-mer suite + full size), C3 / C4 cover times
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kmer.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r04_call9.log
echo "== C3 1e6 reads" | tee -a gpurun_out/r04_call9.log
timeout 300 python tools/bench_kmer.py --reads 1000000 --steps 3 2>&1 | tail -1 | tee -a gpurun_out/r04_call9.log
echo "== C4 1e6 reads (short-read set, trim + split 500)" | tee -a gpurun_out/r04_call9.log
timeout 600 python tools/bench_kmer.py --reads 1000000 --steps 3 --trim-split --short-reads 2>&1 | tail -1 | tee -a gpurun_out/r04_call9.log
