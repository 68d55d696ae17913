#!/bin/bash
# round 4, call 13: seeds per span / tail threshold of the locus path, C3 and C4 cover times
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/r04_call13.log
for v in base s2 s8 s6t2; do
  if [ $v = base ]; then unset FLX_LIB_PATH; else export FLX_LIB_PATH=$PWD/filtlong_amd/lib/exp/libfiltlong_hip_$v.so; fi
  echo "== $v C3" | tee -a gpurun_out/r04_call13.log
  timeout 300 python tools/bench_kmer.py --reads 1000000 --steps 3 2>&1 | tail -1 | cut -c1-260 | tee -a gpurun_out/r04_call13.log
  echo "== $v C4" | tee -a gpurun_out/r04_call13.log
  timeout 600 python tools/bench_kmer.py --reads 1000000 --steps 3 --trim-split --short-reads 2>&1 | tail -1 | cut -c1-260 | tee -a gpurun_out/r04_call13.log
done
