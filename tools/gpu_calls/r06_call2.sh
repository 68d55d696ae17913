cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kmer.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r06_call2_tests.log
{
  echo "== C3 q"; timeout 600 python tools/bench_kmer.py --reads 10000000 --steps 2
  echo "== C3 w"; FLX_KMER_COVER=w timeout 600 python tools/bench_kmer.py --reads 10000000 --steps 2
  echo "== C4 q"; timeout 600 python tools/bench_kmer.py --reads 10000000 --steps 2 --trim-split --short-reads
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_call2.log
