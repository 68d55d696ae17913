# call 30: what is left of the budget — the reference's own suite replayed and the e2e goldens with the final kernels
cd "$GRAFT_REPO_ROOT"
timeout 110 python -m pytest tests/test_gpu_ref_suite.py tests/test_gpu_kmer.py -q -m gpu -x 2>&1 | tail -2 | cut -c1-300
