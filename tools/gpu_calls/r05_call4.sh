cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 python tools/gpu_calls/r05_dbg_fold.py 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r05_call4_dbg.log
timeout 600 python -m pytest tests/test_gpu_kmer.py -q -m gpu -k "integer_grid or fold_variants or locus_path or cover_kernel_boundaries or path_text" 2>&1 | tail -25 | tee gpurun_out/r05_call4_tests.log
{
echo "== C3 (DPP cover)"; timeout 300 python tools/bench_kmer.py --reads 1000000 --steps 3
echo "== C4 (DPP cover)"; timeout 300 python tools/bench_kmer.py --reads 1000000 --steps 3 --trim-split --short-reads
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r05_call4_bench.log
