set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/stats -o c2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/bench_under_stats.json 2> $R/gpurun_out/prof/stats.err
ls -R $R/gpurun_out/prof/stats | head -30
# PMC pass 1: SQ counters
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $R/gpurun_out/prof/pmc_sq -o c2 -- python $R/bench.py --reads 2000000 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof/pmc_sq.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/prof/pmc_lds -o c2 -- python $R/bench.py --reads 2000000 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof/pmc_lds.err
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof/pmc_fetch -o c2 -- python $R/bench.py --reads 2000000 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof/pmc_write -o c2 -- python $R/bench.py --reads 2000000 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof/pmc_write.err
find $R/gpurun_out/prof -name "*.csv" | head -20
du -sh $R/gpurun_out/prof
