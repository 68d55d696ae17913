#!/bin/bash
# round 4, call 14: the path text at order 24 — parity, C4 cover time against order 16
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kmer.py -x -q -m gpu 2>&1 | tail -12 | tee gpurun_out/r04_call14.log
for order in 24 16; do
  echo "== C4 1e6 reads (short-read set, trim + split 500) FLX_KMER_TEXT_ORDER=$order" | tee -a gpurun_out/r04_call14.log
  FLX_KMER_TEXT_ORDER=$order timeout 600 python tools/bench_kmer.py --reads 1000000 --steps 3 --trim-split --short-reads 2>&1 | tail -1 | tee -a gpurun_out/r04_call14.log
done
