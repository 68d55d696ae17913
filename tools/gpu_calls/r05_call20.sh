cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 170 python -m pytest tests/test_gpu_cli.py -q -m gpu -x -k "gzip_input_streamed" 2>&1 | tail -4 | tee gpurun_out/r05_call20.log
