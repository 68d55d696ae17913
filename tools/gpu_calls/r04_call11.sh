#!/bin/bash
# round 4, call 11: the dual-slot Phred kernel — parity, then the window-size sweep (ms per 1e10 bases) against the shipped kernels
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_phred.py -x -q -m gpu -k "dual" 2>&1 | tail -12 | tee gpurun_out/r04_call11.log
for ws in 250 600 640 800 1000 1008 1500 2000 3000; do
  for k in default dual; do
    if [ $k = dual ]; then export FLX_PHRED_KERNEL=dual; else unset FLX_PHRED_KERNEL; fi
    echo -n "$k: " | tee -a gpurun_out/r04_call11.log
    timeout 200 python tools/bench_phred_kernel.py 1000000 $ws 2>&1 | tail -1 | tee -a gpurun_out/r04_call11.log
  done
done
