#!/bin/bash
# the whole -m gpu suite + the default bench line (no profiler, no end-to-end runs): gpurun_out/final/
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2 | cut -c1-300
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; head -c 400 $OUT/bench_default.json; echo
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
