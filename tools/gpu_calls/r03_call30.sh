#!/bin/bash
# round 3: register-history Phred kernel up to ws 1007 (rings in VGPRs + AGPRs) — tests, sweep, and the C2 step again (the unroll flags
# now apply to every instantiation)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_phred.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | cut -c1-300 | tail -4
for ws in 250 511 512 600 640 700 768 900 1000 1007 1008; do
timeout 200 python tools/bench_phred_kernel.py 1000000 $ws 2>&1 | tail -1 | sed -E 's/profile 0 reads 1000000 bases [0-9]+ //'
done
timeout 600 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms_per_step'])"
