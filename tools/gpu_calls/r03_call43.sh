#!/bin/bash
# folds with fma(bit, delta, w): parity + C3/C4 timing
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kmer.py -x -q -m gpu 2>&1 | tail -5
for c in c3 c4; do timeout 600 python bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$c', r['value'], r['ms_per_step'], r['stage_ms_per_step'])"; done
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -5
