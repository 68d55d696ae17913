#!/bin/bash
# round 4, call 20: dual-slot kernel with aligned trailing halves (2-slot ring, 9 waves per CU): parity + sweep
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_phred.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r04_call20.log
for ws in 624 640 656 672 1000 1500 3000; do
  timeout 200 python tools/bench_phred_kernel.py 1000000 $ws 2>&1 | tail -1 | tee -a gpurun_out/r04_call20.log
done
