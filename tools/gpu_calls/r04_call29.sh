# call 29: the cover kernel as shipped (mask fast path, no far-first mode with a text) — parity gate, PMC passes at 1e7 reads, bench line, smoke, fuzz against the reference binary
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/final
mkdir -p $OUT $R/gpurun_out/c29
cd $R
timeout 500 python -m pytest tests/test_gpu_kmer.py -x -q -m gpu > $R/gpurun_out/c29/kmer_tests.log 2>&1
rc=$?; tail -3 $R/gpurun_out/c29/kmer_tests.log | cut -c1-400
[ $rc -ne 0 ] && { echo "GATE: k-mer tests failed"; exit 1; }
timeout 300 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "kmer_mode_properties and mid" 2>&1 | tail -2 | cut -c1-300
rm -rf $R/gpurun_out/prof_kmer; bash tools/prof_kmer.sh 10000000 "c3 c4" light > $OUT/prof_kmer.out 2>&1; grep -E "TCC_|cover" $OUT/prof_kmer.out | head -20
python tools/make_profile_json.py r04 10000000 kmer-only
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<'PY'
import json, os
R = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
j = json.loads(open(R + "/gpurun_out/final/bench_default.json").read().strip().splitlines()[-1])
for c in ("c3", "c4"):
    e = j["extras"][c]; print(c, e["value"], e["ms_per_step"], e["stage_ms_per_step"], e["roofline"].get("traffic"), e["set_build_s_device"], e["cut"])
print("c2", j["value"], j["roofline"]["frac"])
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 200 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_cli.py -x -q -m gpu -k "random_invocations_match or byte_for_byte or kmer" 2>&1 | tail -2 | cut -c1-300
