#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_phred.py -x -q 2>&1 | tail -4 | cut -c1-400
FLX_PHRED_TABLES=private timeout 300 python tools/bench_phred_kernel.py 3000000 2>&1 | tail -1
timeout 300 python tools/bench_phred_kernel.py 3000000 250 1 2>&1 | tail -1
timeout 300 python tools/bench_phred_kernel.py 3000000 250 0 2>&1 | tail -1
timeout 600 python bench.py --config c2wide --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_kernel_ms'], d['roofline']['kernel'])"
