# call 27: the cover kernel with fewer vector instructions (codes by table, scalar text index, RC window for the prefilter, spans the text settles skipped) — parity gate, timing, PMC passes at 1e7 reads, bench line
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/final
mkdir -p $OUT $R/gpurun_out/c27
cd $R
timeout 500 python -m pytest tests/test_gpu_kmer.py -x -q -m gpu > $R/gpurun_out/c27/kmer_tests.log 2>&1
rc=$?; tail -3 $R/gpurun_out/c27/kmer_tests.log | cut -c1-400
[ $rc -ne 0 ] && { echo "GATE: k-mer tests failed"; exit 1; }
for cfg in c3; do for v in 1; do
  FLX_KMER_SAFE1=$v timeout 200 python bench.py --config $cfg --reads 10000000 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/c27/ab_${cfg}_$v.json 2> $R/gpurun_out/c27/ab_${cfg}_$v.err
  python - $R/gpurun_out/c27/ab_${cfg}_$v.json $cfg $v <<'PY'
import re, sys
t = open(sys.argv[1]).read()
m = re.search(r'"cover_kernel": ([0-9.]+)', t); k = re.search(r'"kept_bases": (\d+)', t)
print(sys.argv[2], "SAFE1=" + sys.argv[3], "cover_kernel ms/step", m and m.group(1), "kept_bases", k and k.group(1))
PY
done; done
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
