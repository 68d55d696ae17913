#!/bin/bash
# round 4, call 2: the locus path — parity (k-mer suite incl. test_locus_path_vs_oracle), then cover kernel time with / without it
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kmer.py -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/r04_call2.log
for locus in 1 0; do
  echo "== C3 1e6 reads FLX_KMER_LOCUS=$locus" | tee -a gpurun_out/r04_call2.log
  FLX_KMER_LOCUS=$locus timeout 300 python tools/bench_kmer.py --reads 1000000 --steps 3 2>&1 | tail -1 | tee -a gpurun_out/r04_call2.log
done
echo "== C4-style flags on the assembly set (trim + split 500)" | tee -a gpurun_out/r04_call2.log
timeout 300 python tools/bench_kmer.py --reads 1000000 --steps 3 --trim-split 2>&1 | tail -1 | tee -a gpurun_out/r04_call2.log
