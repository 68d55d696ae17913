# call 24: S1 (single-substitution windows refuted from the text) — parity gate, A/B, PMC passes at 1e7 reads, bench line, full-size cross-check
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/final
mkdir -p $OUT $R/gpurun_out/c24
cd $R
timeout 500 python -m pytest tests/test_gpu_kmer.py -x -q -m gpu > $R/gpurun_out/c24/kmer_tests.log 2>&1
rc=$?; tail -5 $R/gpurun_out/c24/kmer_tests.log | cut -c1-400
[ $rc -ne 0 ] && { echo "GATE: k-mer tests failed"; exit 1; }
for cfg in c3 c4; do for v in 1 0; do
  FLX_KMER_SAFE1=$v timeout 200 python bench.py --config $cfg --reads 10000000 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/c24/ab_${cfg}_$v.json 2> $R/gpurun_out/c24/ab_${cfg}_$v.err
  python - $R/gpurun_out/c24/ab_${cfg}_$v.json $cfg $v <<'PY'
import re, sys
t = open(sys.argv[1]).read()
m = re.search(r'"cover_kernel": ([0-9.]+)', t); k = re.search(r'"kept_bases": (\d+)', t); b = re.search(r'"set_build_s_device": ([0-9.]+)', t)
print(sys.argv[2], "SAFE1=" + sys.argv[3], "cover_kernel ms/step", m and m.group(1), "kept_bases", k and k.group(1), "set build s", b and b.group(1))
PY
done; done
python - <<'PY'
import re, sys, os
R = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
def ms(cfg, v):
    m = re.search(r'"cover_kernel": ([0-9.]+)', open("%s/gpurun_out/c24/ab_%s_%d.json" % (R, cfg, v)).read())
    return float(m.group(1))
gain = ms("c3", 0) / ms("c3", 1)
print("GATE: C3 cover kernel without / with S1 = %.3f" % gain)
sys.exit(0 if gain > 1.03 else 1)
PY
[ $? -ne 0 ] && { echo "GATE: not faster, stopping"; exit 1; }
timeout 300 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "kmer_mode_properties and mid" 2>&1 | tail -2 | cut -c1-300
rm -rf $R/gpurun_out/prof_kmer; bash tools/prof_kmer.sh 10000000 "c3 c4" light > $OUT/prof_kmer.out 2>&1; grep -E "TCC_|cover" $OUT/prof_kmer.out | head -20
python tools/make_profile_json.py r04 10000000 kmer-only
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 300 $OUT/bench_default.json; echo
python - <<'PY'
import json, os
R = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
j = json.loads(open(R + "/gpurun_out/final/bench_default.json").read().strip().splitlines()[-1])
for c in ("c3", "c4"):
    e = j["extras"][c]; print(c, e["value"], e["ms_per_step"], e["stage_ms_per_step"], e["roofline"].get("traffic"), e["cut"])
print("c2", j["value"], j["roofline"]["frac"])
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 400 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "kmer_mode_properties and full" 2>&1 | tail -2 | cut -c1-300
