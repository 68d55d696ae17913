# first call of round 5 (prepared at the end of round 4, when the GPU budget was spent): the multi-rank paths with every rank indexing
# only its byte range (FLX_CLI_RANK_RANGES=1) — the ranks fuzz against the reference binary, the CLI's and the
# communicator's multi-rank tests — then the whole GPU suite as it ships.  If the first block is green: make the range path the
# default in filtlong_amd/cli/main.cpp (rank_ranges: drop the getenv, keep FLX_CLI_RANK_RANGES=0 as the switch back).
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
FLX_CLI_RANK_RANGES=1 timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_cli.py tests/test_gpu_comm2.py -q -m gpu -k "ranks or gpus or sinks or verbose or launch" 2>&1 | tail -6 | tee gpurun_out/r05_call1_ranges.log
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r05_call1_suite.log
