cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash tools/prof_kmer.sh 10000000 "c3" > gpurun_out/r06_call20_prof.log 2>&1
