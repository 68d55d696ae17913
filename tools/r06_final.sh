#!/bin/bash
# round-6 evidence: the full GPU suite, the default bench line (C2 + extras incl. the end-to-end run measured there), the
# rocprofv3 kernel stats of the same command, PMC passes of the C2 kernel and of the k-mer cover kernel, the smoke test.
# Output under gpurun_out/final/.   usage: tools/r06_final.sh [skip-tests]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd $R
if [ "$1" != "skip-tests" ]; then
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log | cut -c1-300
fi
# PMC passes of the k-mer cover kernel first, and the request json on the box: the bench line quotes it (source hash checked)
rm -rf $R/gpurun_out/prof_kmer; bash tools/prof_kmer.sh 10000000 "c3 c4" > $OUT/prof_kmer.out 2>&1; grep -E "TCC_|cover|fold" $OUT/prof_kmer.out | head -30
# ... and of C3 on the two read profiles of round 6 (the passes the request model is made of)
bash tools/prof_kmer.sh 10000000 "c3_indels c3_unrelated" light > $OUT/prof_kmer_profiles.out 2>&1; grep -E "TCC_|cover" $OUT/prof_kmer_profiles.out | head -30
python tools/make_profile_json.py r06 10000000 kmer-only
t0=$SECONDS; timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "python bench.py (default flags, all extras): $((SECONDS - t0)) s wall" | tee $OUT/bench_wall.txt; tail -c 900 $OUT/bench_default.json | head -c 600; echo
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o c2 -- python $R/bench.py --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
python $R/tools/rocprof_summary.py $OUT/stats/c2_results.db > $OUT/c2_kernel_stats.txt 2>&1; head -8 $OUT/c2_kernel_stats.txt | cut -c1-140
rm -rf $OUT/stats
# PMC passes of one C2 launch (separate runs, counters only)
B="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_lds -o p -- $B > /dev/null 2> $OUT/pmc_lds.err
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $B > /dev/null 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $B > /dev/null 2> $OUT/pmc_write.err
for d in pmc_lds pmc_fetch pmc_write; do python $R/tools/rocprof_summary.py $OUT/$d/p_results.db phred > $OUT/$d.txt 2>&1; rm -rf $OUT/$d; done
cat $OUT/pmc_fetch.txt $OUT/pmc_write.txt | grep -E "FETCH|WRITE"
cd $R
python tools/make_profile_json.py r06 10000000
mkdir -p $OUT/profiles; cp profiles/r06_* $OUT/profiles/
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
