cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for prof in 0 1 2; do
  echo "== C3 profile $prof q"; timeout 600 python tools/bench_kmer.py --reads 2000000 --steps 3 --profile $prof
done
echo "== C3 10M q"; timeout 600 python tools/bench_kmer.py --reads 10000000 --steps 2
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_call21.log
timeout 1500 python -m pytest tests/test_gpu_kmer.py "tests/test_gpu_fullsize.py::test_kmer_read_profiles_whole_population" -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r06_call21_tests.log
