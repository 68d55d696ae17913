# round 5, call 18: the whole GPU suite, smoke() and the default bench line on the final tree
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/final2
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/final2/pytest_gpu.log 2>&1; tail -4 gpurun_out/final2/pytest_gpu.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/final2/bench_default.json 2> gpurun_out/final2/bench_default.err; tail -c 600 gpurun_out/final2/bench_default.json; echo
