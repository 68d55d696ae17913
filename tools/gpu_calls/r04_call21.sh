#!/bin/bash
# round 4, call 21: after the last Phred kernel change — fuzz (window sizes 600 / 1500 through the CLI), C2 stats + PMC passes again, the bench line
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_cli.py -q -m gpu -k "random_invocations_match or byte_for_byte" 2>&1 | tail -3
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 300 $OUT/bench_default.json; echo
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o c2 -- python $R/bench.py --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
python $R/tools/rocprof_summary.py $OUT/stats/c2_results.db > $OUT/c2_kernel_stats.txt 2>&1; head -3 $OUT/c2_kernel_stats.txt | cut -c1-140
rm -rf $OUT/stats
B="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_lds -o p -- $B > /dev/null 2> $OUT/pmc_lds.err
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $B > /dev/null 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $B > /dev/null 2> $OUT/pmc_write.err
for d in pmc_lds pmc_fetch pmc_write; do python $R/tools/rocprof_summary.py $OUT/$d/p_results.db phred > $OUT/$d.txt 2>&1; rm -rf $OUT/$d; done
cat $OUT/pmc_fetch.txt $OUT/pmc_write.txt | grep -E "FETCH|WRITE"
cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
