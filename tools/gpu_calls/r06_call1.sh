cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
  echo "== stats C3"; FLX_LIB_PATH=filtlong_amd/lib/exp/libfiltlong_hip_stats.so timeout 600 python tools/exp/cover_stats.py 1000000
  echo "== stats C4"; FLX_LIB_PATH=filtlong_amd/lib/exp/libfiltlong_hip_stats.so timeout 600 python tools/exp/cover_stats.py 1000000 --short-reads
  echo "== C3 default"; timeout 600 python tools/bench_kmer.py --reads 10000000 --steps 2
  echo "== C4 default"; timeout 600 python tools/bench_kmer.py --reads 10000000 --steps 2 --trim-split --short-reads
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_call1.log
