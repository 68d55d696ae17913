# call 28: compile-time variants of the cover kernel (waves per SIMD, no far-first mode, mask fast path, seeds per span), C3 at 3e6 reads
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c28
FLX_LIB_PATH=$PWD/filtlong_amd/lib/exp/libflx_vfff7.so timeout 200 python -m pytest tests/test_gpu_kmer.py -x -q -m gpu 2>&1 | tail -2 | cut -c1-200
FLX_LIB_PATH=$PWD/filtlong_amd/lib/exp/libflx_vfff7.so timeout 200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "kmer_mode_properties and mid" 2>&1 | tail -1 | cut -c1-200
for v in base w7 w6 ff vf vfff vfff7 vfff6 s2; do
  if [ $v = base ]; then unset FLX_LIB_PATH; else export FLX_LIB_PATH=$PWD/filtlong_amd/lib/exp/libflx_$v.so; fi
  timeout 120 python bench.py --config c3 --reads 3000000 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/c28/$v.json 2> gpurun_out/c28/$v.err
  python - gpurun_out/c28/$v.json $v <<'PY'
import re, sys
t = open(sys.argv[1]).read()
m = re.search(r'"cover_kernel": ([0-9.]+)', t); k = re.search(r'"kept_bases": (\d+)', t)
print(sys.argv[2], "cover_kernel ms/step", m and m.group(1), "kept_bases", k and k.group(1))
PY
done
