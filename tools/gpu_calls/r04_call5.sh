#!/bin/bash
# round 4, call 5: witness-guided pairing of the path text — parity, C4 cover time
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kmer.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r04_call5.log
for locus in 1 0; do
  echo "== C4 1e6 reads (short-read set, trim + split 500) FLX_KMER_LOCUS=$locus" | tee -a gpurun_out/r04_call5.log
  FLX_KMER_LOCUS=$locus timeout 600 python tools/bench_kmer.py --reads 1000000 --steps 3 --trim-split --short-reads 2>&1 | tail -1 | tee -a gpurun_out/r04_call5.log
done
