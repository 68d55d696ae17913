cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 python tools/gpu_calls/r05_dbg_fold.py 2>&1 | grep -v "amdgpu.ids" | cut -c1-300 | tee gpurun_out/r05_call5_dbg.log
timeout 900 python -m pytest tests/test_gpu_kmer.py -q -m gpu 2>&1 | tail -5 | tee -a gpurun_out/r05_call5_dbg.log
{
for G in 1 0; do
  echo "== C3 FLX_KMER_FOLD_GRID=$G"; FLX_KMER_FOLD_GRID=$G timeout 300 python tools/bench_kmer.py --reads 2000000 --steps 3
  echo "== C4 FLX_KMER_FOLD_GRID=$G"; FLX_KMER_FOLD_GRID=$G timeout 300 python tools/bench_kmer.py --reads 2000000 --steps 3 --trim-split --short-reads
done
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r05_call5_bench.log
