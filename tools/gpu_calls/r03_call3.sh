#!/bin/bash
# round 3: the wave-level cover kernel — k-mer tests, then C3 at 1e6 reads with the new and the old kernel
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kmer.py -x -q 2>&1 | tail -5 | cut -c1-600
for v in w v2; do
  FLX_KMER_COVER=$v timeout 300 python bench.py --config c3 --reads 1000000 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['cut'])"
done
FLX_KMER_COVER=w timeout 300 python bench.py --config c4 --reads 1000000 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('c4', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['cut'])"
