#!/bin/bash
# round 3: ablations of the wave-level cover kernel (C3, 1e6 reads) and the BASELINE-size k-mer tests (reads2 + oracle global stage)
cd $GRAFT_REPO_ROOT
for v in nofar nolookups; do
FLX_LIB_PATH=filtlong_amd/lib/exp/libfiltlong_hip_$v.so timeout 300 python bench.py --config c3 --reads 1000000 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', d['value'], d['ms_per_step'], d['stage_ms_per_step']['cover_kernel'])"
done
timeout 2400 python -m pytest tests/test_gpu_fullsize.py -x -q 2>&1 | tail -5 | cut -c1-800
