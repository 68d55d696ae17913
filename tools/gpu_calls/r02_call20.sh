#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_rank.py tests/test_gpu_sharded.py tests/test_gpu_comm2.py -x -q 2>&1 | tail -3 | cut -c1-400
timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms_per_step'])"
