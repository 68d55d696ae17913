#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_sharded.py tests/test_gpu_rank.py -x -q 2>&1 | tail -8 | cut -c1-600
timeout 900 tools/bench_e2e_big.sh 1000000 2>&1 | tr '\r' '\n' | grep -v " reads (" | tail -50 | cut -c1-400
