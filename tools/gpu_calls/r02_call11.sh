#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_cli.py tests/test_gpu_ref_suite.py -x -q 2>&1 | tail -15 | cut -c1-600
timeout 600 tools/bench_e2e_gz.sh 150000 2>&1 | tr '\r' '\n' | grep -v " reads (" | tail -45 | cut -c1-300
