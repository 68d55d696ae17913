#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_rank.py tests/test_gpu_sharded.py -x -q 2>&1 | tail -8 | cut -c1-600
timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/bench_walk.json 2> gpurun_out/bench_walk.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_walk.json").read().strip().split("\n")[-1])
print({k: d[k] for k in ("value", "ms_per_step") if k in d})
print(d.get("roofline")); print(d.get("stage_ms_per_step"))
PY
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_walk -o walk -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof_walk/walk_results.db 2>&1 | head -30 | cut -c1-160
