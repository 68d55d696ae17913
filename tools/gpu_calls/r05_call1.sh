# round 5, call 1: the multi-rank paths with every rank indexing only its byte range (FLX_CLI_RANK_RANGES=1) —
# the ranks fuzz against the reference binary, the CLI's and the communicator's multi-rank tests.  Green => the range path becomes the default.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
FLX_CLI_RANK_RANGES=1 timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_cli.py tests/test_gpu_comm2.py -q -m gpu -k "ranks or gpus or sinks or verbose or launch" 2>&1 | tail -15 | tee gpurun_out/r05_call1_ranges.log
