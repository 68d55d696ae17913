cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for V in noslow; do
  export FLX_LIB_PATH=$PWD/filtlong_amd/lib/exp/libfiltlong_hip_$V.so
  echo "== C3 $V"; timeout 300 python tools/bench_kmer.py --reads 2000000 --steps 3
  echo "== C4 $V"; timeout 300 python tools/bench_kmer.py --reads 2000000 --steps 3 --trim-split --short-reads
done
unset FLX_LIB_PATH
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_fold -o fold -- python $GRAFT_REPO_ROOT/tools/bench_kmer.py --reads 2000000 --steps 2 --trim-split --short-reads > /dev/null 2>&1
find /tmp/prof_fold -name "*kernel_stats*" | head -2 | xargs -I{} sh -c 'head -30 {}'
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $GRAFT_REPO_ROOT/gpurun_out/r05_call7_bench.log
