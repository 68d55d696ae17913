cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_kmer.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -12 | tee gpurun_out/r06_call37_tests.log
