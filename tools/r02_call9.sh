#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_phred.py -x -q -m gpu 2>&1 | tail -3
{
for prof in 0 1; do
echo "== regs plain profile $prof"; timeout 200 python tools/bench_phred_kernel.py 3000000 250 $prof 2>&1 | grep reads
echo "== regs private profile $prof"; FLX_PHRED_TABLES=private timeout 200 python tools/bench_phred_kernel.py 3000000 250 $prof 2>&1 | grep reads
echo "== ring profile $prof"; FLX_PHRED_KERNEL=ring timeout 200 python tools/bench_phred_kernel.py 3000000 250 $prof 2>&1 | grep reads
done
} 2>&1 | tee $OUT/profiles_cmp.log
