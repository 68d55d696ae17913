#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kmer.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | cut -c1-300 | tail -5
for c in c3 c4; do
timeout 300 python bench.py --config $c --reads 1000000 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$c', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['cut']['kept_bases'], d['set_build_s_device'])"
done
bash tools/prof_kmer.sh 1000000 c3 2>&1 | grep -E "TCC_|FETCH" | grep cover
