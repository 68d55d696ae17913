#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "" ct64 ct128 ct512; do
L=filtlong_amd/lib/libfiltlong_hip.so; [ -n "$v" ] && L=filtlong_amd/lib/exp/libfiltlong_hip_$v.so
FLX_LIB_PATH=$L timeout 300 python bench.py --config c3 --reads 1000000 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('c3 ${v:-ct256}', d['value'], d['ms_per_step'], d['stage_ms_per_step']['cover_kernel'], d['cut']['kept_bases'])"
done
