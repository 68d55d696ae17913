#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_phred.py -x -q -m gpu > $OUT/t_phred.log 2>&1; tail -5 $OUT/t_phred.log
{
echo "== default 3M"; timeout 200 python tools/bench_phred_kernel.py 3000000 250 2>&1 | grep reads
echo "== fold4 3M"; FLX_LIB_PATH=$R/filtlong_amd/lib/exp/libfiltlong_hip_fold4.so timeout 200 python tools/bench_phred_kernel.py 3000000 250 2>&1 | grep reads
echo "== private 3M"; FLX_PHRED_TABLES=private timeout 200 python tools/bench_phred_kernel.py 3000000 250 2>&1 | grep reads
echo "== default 10M"; timeout 300 python tools/bench_phred_kernel.py 10000000 250 2>&1 | grep reads
echo "== private 10M"; FLX_PHRED_TABLES=private timeout 300 python tools/bench_phred_kernel.py 10000000 250 2>&1 | grep reads
} > $OUT/variants.log 2>&1
cat $OUT/variants.log
cd /tmp
B="python $R/tools/bench_phred_kernel.py 3000000 250"
timeout 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $B > /dev/null 2> $OUT/pmc_fetch.err
python $R/tools/rocprof_summary.py $OUT/pmc_fetch/p_results.db phred > $OUT/pmc_fetch.txt 2>&1
rm -rf $OUT/pmc_fetch
cat $OUT/pmc_fetch.txt | grep -v "^$"
