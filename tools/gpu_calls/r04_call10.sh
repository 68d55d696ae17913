#!/bin/bash
# round 4, call 10: the folds by events (review item 5) — parity, then fold time per 1e10 positions with and without, C3 and C4 flags;
# the set without its pair table
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kmer.py -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r04_call10.log
for ev in 0 1; do
  echo "== C3 1e6 reads FLX_KMER_FOLD_EVENTS=$ev" | tee -a gpurun_out/r04_call10.log
  FLX_KMER_FOLD_EVENTS=$ev timeout 300 python tools/bench_kmer.py --reads 1000000 --steps 3 2>&1 | tail -1 | tee -a gpurun_out/r04_call10.log
  echo "== C4 flags on the assembly set FLX_KMER_FOLD_EVENTS=$ev" | tee -a gpurun_out/r04_call10.log
  FLX_KMER_FOLD_EVENTS=$ev FLX_API_TIMING=1 timeout 300 python tools/bench_kmer.py --reads 1000000 --steps 3 --trim-split 2>&1 | tail -1 | tee -a gpurun_out/r04_call10.log
done
