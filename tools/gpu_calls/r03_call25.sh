#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in g2 g4; do
L=filtlong_amd/lib/libfiltlong_hip.so; [ $v = g2 ] && L=filtlong_amd/lib/exp/libfiltlong_hip_g2.so
for c in c3; do
FLX_LIB_PATH=$L timeout 300 python bench.py --config $c --reads 1000000 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$c $v', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['cut']['kept_bases'])"
done
done
done
