cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_comm2.py -q -m gpu -x 2>&1 | tail -5 | tee gpurun_out/r05_call10_tests.log
timeout 900 bash tools/bench_e2e_kmer.sh 1000000 50000000 r05 > gpurun_out/r05_call10_e2ek.out 2>&1; tail -5 gpurun_out/r05_call10_e2ek.out
timeout 900 bash tools/bench_rank_ranges.sh 2000000 8 r05 > gpurun_out/r05_call10_rr.out 2>&1; tail -5 gpurun_out/r05_call10_rr.out
