#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_kmer.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4 | cut -c1-400
for c in c3 c4; do
timeout 600 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$c', d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
done
timeout 300 python bench.py --config c3 --reads 1000000 --fixed-len 500 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('c3 500bp', d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
