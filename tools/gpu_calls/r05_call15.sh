cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 bash tools/bench_e2e_kmer.sh 1000000 50000000 r05 > gpurun_out/r05_call15_e2ek.out 2>&1; tail -4 gpurun_out/r05_call15_e2ek.out
timeout 600 bash tools/bench_e2e_big.sh 1000000 r05 > gpurun_out/r05_call15_big.out 2>&1; tail -3 gpurun_out/r05_call15_big.out
timeout 600 bash tools/bench_e2e_gz.sh 150000 r05 > gpurun_out/r05_call15_gz.out 2>&1; tail -3 gpurun_out/r05_call15_gz.out
timeout 900 bash tools/bench_rank_ranges.sh 2000000 8 r05 > gpurun_out/r05_call15_rr.out 2>&1; tail -3 gpurun_out/r05_call15_rr.out
