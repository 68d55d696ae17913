#!/bin/bash
# round 4, call 6: multi-rank control flow (rank-0-only exact fallback + broadcast, bench self-check), then the default bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_comm2.py tests/test_gpu_rank.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r04_call6.log
timeout 900 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; echo "bench rc $?" | tee -a gpurun_out/r04_call6.log
python - <<'PY' | tee -a gpurun_out/r04_call6.log
import json
j=json.loads(open('gpurun_out/r04_bench_default.json').read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j.get("cpu_baseline",{}).get("value"))
for k in ("c3","c4"):
    e=j["extras"][k]; print(k, e.get("value"), e.get("ms_per_step"), e.get("stage_ms_per_step"), e.get("roofline",{}).get("frac"), e.get("error"))
print(j["extras"].get("end_to_end_cli"))
PY
