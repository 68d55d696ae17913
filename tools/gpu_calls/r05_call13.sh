cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kmer.py tests/test_gpu_fullsize.py -q -m gpu -k "not c2_full" 2>&1 | tail -5 | tee gpurun_out/r05_call13_tests.log
{
  echo "== C3 default"; timeout 600 python tools/bench_kmer.py --reads 10000000 --steps 2
  echo "== C4 default"; timeout 600 python tools/bench_kmer.py --reads 10000000 --steps 2 --trim-split --short-reads
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r05_call13_bench.log
