# round 5, call 3: the window folds on the integer grid — against the oracle and the FP kernel, then what they cost (C3 / C4 at 1e6 reads)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kmer.py -q -m gpu -x -k "integer_grid or fold_variants or long_reads_vs_oracle" 2>&1 | tail -25 | tee gpurun_out/r05_call3_tests.log
{
for G in 1 0; do
  echo "== C3 FLX_KMER_FOLD_GRID=$G"; FLX_KMER_FOLD_GRID=$G timeout 300 python tools/bench_kmer.py --reads 1000000 --steps 3
  echo "== C4 FLX_KMER_FOLD_GRID=$G"; FLX_KMER_FOLD_GRID=$G timeout 300 python tools/bench_kmer.py --reads 1000000 --steps 3 --trim-split --short-reads
done
} 2>&1 | grep -v Warning | tee gpurun_out/r05_call3_bench.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "mid" 2>&1 | tail -15 | tee gpurun_out/r05_call3_mid.log
