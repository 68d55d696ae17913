cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1700 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -25 | tee gpurun_out/r05_call12_suite.log
