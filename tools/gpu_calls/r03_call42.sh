#!/bin/bash
# per-kernel times of C4 (1e6 reads) with the one-lane-per-child folds
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/c42; mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/k -o p -- python $R/bench.py --config c4 --reads 1000000 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2>/dev/null
python $R/tools/rocprof_summary.py $OUT/k/p_results.db kmer > $OUT/k_c4.txt 2>&1; rm -rf $OUT/k
cat $OUT/k_c4.txt | cut -c1-200
