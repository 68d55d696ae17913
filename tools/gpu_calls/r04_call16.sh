#!/bin/bash
# round 4, call 16: cover kernel at 8 waves per SIMD (63 VGPRs) against 7 (68)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/r04_call16.log
for v in base t64 t128 t512; do
  if [ $v = base ]; then unset FLX_LIB_PATH; else export FLX_LIB_PATH=$PWD/filtlong_amd/lib/exp/libfiltlong_hip_$v.so; fi
  echo "== $v C3" | tee -a gpurun_out/r04_call16.log
  timeout 300 python tools/bench_kmer.py --reads 1000000 --steps 3 2>&1 | tail -1 | cut -c1-260 | tee -a gpurun_out/r04_call16.log
done
