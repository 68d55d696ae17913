#!/bin/bash
# round-end evidence: the full GPU suite, the default bench line, the rocprofv3 kernel stats of the same command, both
# end-to-end runs.  Output under gpurun_out/final/.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log | cut -c1-300
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default.json | head -c 300; echo
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o c2 -- python $R/bench.py --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
python $R/tools/rocprof_summary.py $OUT/stats/c2_results.db > $OUT/c2_kernel_stats.txt 2>&1; head -8 $OUT/c2_kernel_stats.txt | cut -c1-140
rm -rf $OUT/stats
cd $R
timeout 900 tools/bench_e2e_big.sh 1000000 > $OUT/e2e_big.out 2>&1; tail -14 $OUT/e2e_big.out | cut -c1-200
timeout 600 tools/bench_e2e_gz.sh 150000 > $OUT/e2e_gz.out 2>&1; tail -16 $OUT/e2e_gz.out | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
