#!/bin/bash
cd $GRAFT_REPO_ROOT
export FLX_LIB_PATH=$GRAFT_REPO_ROOT/filtlong_amd/lib/exp/libfiltlong_hip_mfma.so
timeout 900 python -m pytest tests/test_gpu_phred.py -x -q 2>&1 | tail -4 | cut -c1-400
timeout 300 python tools/bench_phred_kernel.py 3000000 2>&1 | tail -1
FLX_PHRED_TABLES=private timeout 300 python tools/bench_phred_kernel.py 3000000 2>&1 | tail -1
unset FLX_LIB_PATH
timeout 300 python tools/bench_phred_kernel.py 3000000 2>&1 | tail -1
FLX_PHRED_TABLES=private timeout 300 python tools/bench_phred_kernel.py 3000000 2>&1 | tail -1
