#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02
cd /tmp
timeout 120 $R/tools/lanestream 2000000 10240 2>&1 | tee $OUT/lanestream.txt
timeout 200 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/ls1 -o p -- $R/tools/lanestream 2000000 10240 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $OUT/ls2 -o p -- $R/tools/lanestream 2000000 10240 > /dev/null 2>&1
python $R/tools/rocprof_summary.py $OUT/ls1/p_results.db > $OUT/lanestream_pmc1.txt 2>&1
python $R/tools/rocprof_summary.py $OUT/ls2/p_results.db > $OUT/lanestream_pmc2.txt 2>&1
rm -rf $OUT/ls1 $OUT/ls2
cat $OUT/lanestream_pmc1.txt $OUT/lanestream_pmc2.txt | grep -v "^$"
