#!/bin/bash
# round 3: register-history Phred kernel for window sizes 1..511 — all Phred tests, then the window-size sweep (1e6 reads, kernel alone)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_phred.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | cut -c1-300
for ws in 8 16 40 47 48 250 319 320 400 500 511 512 1000; do
timeout 200 python tools/bench_phred_kernel.py 1000000 $ws 2>&1 | tail -1 | sed -E 's/profile 0 reads 1000000 //'
done
