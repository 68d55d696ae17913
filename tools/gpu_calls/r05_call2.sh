# round 5, call 2: the whole GPU suite on the tree with the streamed reference reader, raw stderr comparisons, rank ranges by default,
# whole-population parity at full size
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
nproc > gpurun_out/r05_call2_nproc.txt; free -g | head -2 >> gpurun_out/r05_call2_nproc.txt
timeout 1700 python -m pytest tests -q -m gpu --durations=15 2>&1 | tail -40 | tee gpurun_out/r05_call2_suite.log
