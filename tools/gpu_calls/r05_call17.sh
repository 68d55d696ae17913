cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cli.py -q -m gpu -x -k "gzip_input_streamed or sinks or rank_that_cannot or verbose_matches or stderr_is_the_references" 2>&1 | tail -25 | tee gpurun_out/r05_call17_cli.log
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_comm2.py -q -m gpu -x -k "ranks or gpus or launch or damaged" 2>&1 | tail -15 | tee gpurun_out/r05_call17_fuzz.log
