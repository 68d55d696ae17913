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
echo "== default 10M"; timeout 300 python tools/bench_phred_kernel.py 10000000 250 2>&1 | grep reads
} > $OUT/variants.log 2>&1
cat $OUT/variants.log
cd /tmp
B="python $R/tools/bench_phred_kernel.py 3000000 250"
timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU -d $OUT/p1 -o p -- $B > /dev/null 2> $OUT/p1.err
FLX_LIB_PATH=$R/filtlong_amd/lib/exp/libfiltlong_hip_nodma.so timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU -d $OUT/p2 -o p -- $B > /dev/null 2> $OUT/p2.err
python $R/tools/rocprof_summary.py $OUT/p1/p_results.db phred > $OUT/pmc_default.txt 2>&1
python $R/tools/rocprof_summary.py $OUT/p2/p_results.db phred > $OUT/pmc_nodma.txt 2>&1
rm -rf $OUT/p1 $OUT/p2
cat $OUT/pmc_default.txt $OUT/pmc_nodma.txt | grep -v "^$"
