#!/bin/bash
# round 4, call 19: the default bench line with the k-mer end-to-end extra
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
( time timeout 900 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err ) 2>&1 | tail -3
python - <<'PY'
import json
j=json.loads(open('gpurun_out/final/bench_default.json').read().strip().splitlines()[-1])
print(j["value"], j["roofline"]["frac"], j["extras"]["c3"]["value"], j["extras"]["c3"]["roofline"].get("request_model",{}) and j["extras"]["c3"]["roofline"]["request_model"]["model_ms"], j["extras"]["c4"]["value"])
print(j["extras"]["end_to_end_cli_kmer"])
PY
