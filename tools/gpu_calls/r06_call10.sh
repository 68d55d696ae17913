cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for prof in 0 1 2; do
  echo "== C3 profile $prof q"; timeout 600 python tools/bench_kmer.py --reads 2000000 --steps 3 --profile $prof
done
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_call10.log
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r06_call10_tests.log
