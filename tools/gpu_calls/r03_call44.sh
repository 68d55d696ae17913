#!/bin/bash
# gzip input end to end with the block-parallel inflater in pass 1
cd $GRAFT_REPO_ROOT
timeout 900 bash tools/bench_e2e_gz.sh 150000 r03 > /dev/null 2>&1
cat gpurun_out/r03_e2e_gz.log; cat gpurun_out/r03_e2e_gz.json
