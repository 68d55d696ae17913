#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_cli.py tests/test_gpu_comm2.py tests/test_gpu_ref_suite.py tests/test_gpu_stream.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | cut -c1-400 | tail -8
