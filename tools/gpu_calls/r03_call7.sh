#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 0 1 2 3; do
FLX_LIB_PATH=filtlong_amd/lib/exp/libfiltlong_hip_far$v.so timeout 300 python bench.py --config c3 --reads 1000000 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('far$v', d['value'], d['ms_per_step'], d['stage_ms_per_step']['cover_kernel'])"
done
done
