#!/bin/bash
# rocprofv3 evidence for the scoring hot path (run on the GPU box through gpurun).  Writes rocpd databases
# and text summaries under gpurun_out/prof/; the summaries are then copied into profiles/.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp
# 1. kernel trace + stats of the SAME command bench.py's default run uses (C2, 10 M reads)
rocprofv3 --kernel-trace --stats -d $OUT/stats -o c2 -- python $R/bench.py --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
python $R/tools/rocprof_summary.py $OUT/stats/c2_results.db > $OUT/c2_kernel_stats.txt
# 2. PMC passes (separate runs, counters only), the full C2 batch, one launch each
B="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/pmc_sq -o p -- $B > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_lds -o p -- $B > /dev/null 2> $OUT/pmc_lds.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $B > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $B > /dev/null 2> $OUT/pmc_write.err
for d in pmc_sq pmc_lds pmc_fetch pmc_write; do python $R/tools/rocprof_summary.py $OUT/$d/p_results.db phred > $OUT/$d.txt; done
# HBM traffic of one flx_score_phred_ring launch, corrected as MI355X_MICROARCH.md §HBM prescribes:
# FETCH_SIZE (KiB) counts exactly half of a wide coalesced streaming read on gfx950 -> x2; WRITE_SIZE as reported.
python - <<PY
import json, re
def val(f, c):
    for l in open(f):
        if c in l and "flx_score_phred" in l:
            return float(l.split()[-2])
fetch = val("$OUT/pmc_fetch.txt", "FETCH_SIZE"); write = val("$OUT/pmc_write.txt", "WRITE_SIZE")
json.dump({"kernel": "flx_score_phred_regs", "workload": "C2 (10M reads, one launch)", "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
           "traffic_bytes": 2 * fetch * 1024 + write * 1024,
           "correction": "read side x2 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B); calibrated on a known byte count with this access pattern: tools/membench under --pmc FETCH_SIZE reports 0.5002 of the bytes (profiles/r02_microbench.txt)"},
          open("$OUT/traffic_c2.json", "w"), indent=1)
PY
rm -rf $OUT/stats $OUT/pmc_sq $OUT/pmc_lds $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT
