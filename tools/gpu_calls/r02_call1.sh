#!/bin/bash
# GPU call 1 of round 2: LDS gather microbenchmarks + FETCH_SIZE calibration on a known byte count
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02
mkdir -p $OUT
cd $R
timeout 300 tools/ldsgather2 > $OUT/ldsgather2.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/calib -o m -- $R/tools/membench 2000000 10240 7 > $OUT/membench_pmc.txt 2>&1
python $R/tools/rocprof_summary.py $OUT/calib/m_results.db > $OUT/membench_fetch.txt 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/calibw -o m -- $R/tools/membench 2000000 10240 7 > /dev/null 2>&1
python $R/tools/rocprof_summary.py $OUT/calibw/m_results.db > $OUT/membench_write.txt 2>&1
rm -rf $OUT/calib $OUT/calibw
tail -50 $OUT/ldsgather2.txt
cat $OUT/membench_fetch.txt
