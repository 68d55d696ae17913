#!/bin/bash
# round 4, call 12: dual-slot kernel with a trailing ring of 3 half slots (7 waves per CU) against 4 (6 waves): parity + sweep
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
FLX_LIB_PATH=$PWD/filtlong_amd/lib/exp/libfiltlong_hip_nh3.so timeout 900 python -m pytest tests/test_gpu_phred.py -x -q -m gpu -k "dual" 2>&1 | tail -3 | tee gpurun_out/r04_call12.log
for ws in 640 1000 1500 3000; do
  for k in nh4 nh3; do
    if [ $k = nh3 ]; then export FLX_LIB_PATH=$PWD/filtlong_amd/lib/exp/libfiltlong_hip_nh3.so; else unset FLX_LIB_PATH; fi
    echo -n "$k: " | tee -a gpurun_out/r04_call12.log
    FLX_PHRED_KERNEL=dual timeout 200 python tools/bench_phred_kernel.py 1000000 $ws 2>&1 | tail -1 | tee -a gpurun_out/r04_call12.log
  done
done
