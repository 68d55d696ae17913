#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kmer.py -x -q 2>&1 | grep -E "passed|failed|Error" | cut -c1-300
for v in "" ff16 ff24 ff40 ff64; do
L=filtlong_amd/lib/libfiltlong_hip.so; [ -n "$v" ] && L=filtlong_amd/lib/exp/libfiltlong_hip_$v.so
FLX_LIB_PATH=$L timeout 300 python bench.py --config c3 --reads 1000000 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('c3 ${v:-ff32}', d['value'], d['ms_per_step'], d['stage_ms_per_step']['cover_kernel'], d['cut']['kept_bases'])"
done
