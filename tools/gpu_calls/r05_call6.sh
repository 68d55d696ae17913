cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for V in default walk1 walk16 fp; do
  unset FLX_LIB_PATH FLX_KMER_FOLD_GRID
  case $V in walk1|walk16) export FLX_LIB_PATH=$PWD/filtlong_amd/lib/exp/libfiltlong_hip_$V.so;; fp) export FLX_KMER_FOLD_GRID=0;; esac
  echo "== C3 $V"; timeout 300 python tools/bench_kmer.py --reads 2000000 --steps 3
  echo "== C4 $V"; timeout 300 python tools/bench_kmer.py --reads 2000000 --steps 3 --trim-split --short-reads
done
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r05_call6_bench.log
