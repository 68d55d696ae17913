#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02
cd $R
for v in nowait halfdma nodma; do echo "== $v 3M"; FLX_LIB_PATH=$R/filtlong_amd/lib/exp/libfiltlong_hip_$v.so timeout 200 python tools/bench_phred_kernel.py 3000000 250 2>&1 | grep reads; done 2>&1 | tee $OUT/variants2.log
