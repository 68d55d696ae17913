# call 23: the CLI after the write-error checks (every sink, ranks, /dev/full), ranks fuzz
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_comm2.py tests/test_gpu_fuzz.py -q -m gpu -k "test_gpu_cli or two_ranks or forked_ranks or launch_fails" 2>&1 | tail -12 | tee gpurun_out/r04_call23.log
