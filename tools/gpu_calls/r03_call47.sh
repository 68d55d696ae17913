#!/bin/bash
# 20 GB end to end (both binaries), then the default bench line once more (request_model now matches the source hash)
cd $GRAFT_REPO_ROOT
timeout 900 bash tools/bench_e2e_big.sh 1000000 r03 > /dev/null 2>&1
tail -25 gpurun_out/r03_e2e_big.log; cat gpurun_out/r03_e2e_big.json
cp gpurun_out/r03_e2e_big.json profiles/ 2>/dev/null
mkdir -p gpurun_out/final
timeout 900 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err; tail -c 600 gpurun_out/final/bench_default.json
