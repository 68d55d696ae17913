#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_phred.py -x -q -m gpu > $OUT/t_phred.log 2>&1; tail -5 $OUT/t_phred.log
{
echo "== default 3M"; timeout 200 python tools/bench_phred_kernel.py 3000000 250 2>&1 | grep reads
echo "== nodma 3M"; FLX_LIB_PATH=$R/filtlong_amd/lib/exp/libfiltlong_hip_nodma.so timeout 200 python tools/bench_phred_kernel.py 3000000 250 2>&1 | grep reads
echo "== private 3M"; FLX_PHRED_TABLES=private timeout 200 python tools/bench_phred_kernel.py 3000000 250 2>&1 | grep reads
echo "== default 10M"; timeout 300 python tools/bench_phred_kernel.py 10000000 250 2>&1 | grep reads
echo "== private 10M"; FLX_PHRED_TABLES=private timeout 300 python tools/bench_phred_kernel.py 10000000 250 2>&1 | grep reads
} > $OUT/variants.log 2>&1
cd /tmp
B="python $R/tools/bench_phred_kernel.py 3000000 250"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $B > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_lds -o p -- $B > /dev/null 2> $OUT/pmc_lds.err
for d in pmc_lds pmc_fetch; do python $R/tools/rocprof_summary.py $OUT/$d/p_results.db phred > $OUT/$d.txt 2>&1; done
rm -rf $OUT/pmc_lds $OUT/pmc_fetch
cat $OUT/variants.log; cat $OUT/pmc_lds.txt $OUT/pmc_fetch.txt | grep -v "^$"
