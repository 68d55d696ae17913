#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02
mkdir -p $OUT
cd $R
{
for v in nodma nomask w16; do echo "== variant $v"; FLX_LIB_PATH=$R/filtlong_amd/lib/exp/libfiltlong_hip_$v.so timeout 200 python tools/bench_phred_kernel.py 3000000 250 2>&1 | grep reads; done
echo "== default 10M"; timeout 300 python tools/bench_phred_kernel.py 10000000 250 2>&1 | grep reads
echo "== ring 10M"; FLX_PHRED_KERNEL=ring timeout 300 python tools/bench_phred_kernel.py 10000000 250 2>&1 | grep reads
} > $OUT/variants.log 2>&1
cd /tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
B="python $R/tools/bench_phred_kernel.py 3000000 250"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/pmc_sq -o p -- $B > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_lds -o p -- $B > /dev/null 2> $OUT/pmc_lds.err
rocprofv3 --pmc SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC -d $OUT/pmc_if -o p -- $B > /dev/null 2> $OUT/pmc_if.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $B > /dev/null 2> $OUT/pmc_fetch.err
for d in pmc_sq pmc_lds pmc_if pmc_fetch; do python $R/tools/rocprof_summary.py $OUT/$d/p_results.db phred > $OUT/$d.txt 2>&1; done
rm -rf $OUT/pmc_sq $OUT/pmc_lds $OUT/pmc_if $OUT/pmc_fetch
cat $OUT/variants.log; cat $OUT/pmc_sq.txt $OUT/pmc_lds.txt $OUT/pmc_if.txt $OUT/pmc_fetch.txt | grep -v "^$"
grep -i -E "icache|ifetch|inst_cache" $OUT/counters.txt | head -20
