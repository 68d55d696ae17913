#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_cli.py tests/test_gpu_ref_suite.py -x -q 2>&1 | tail -15 | cut -c1-400
