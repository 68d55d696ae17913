# call 31: the last minute of the budget — ranks that index only their byte range (FLX_CLI_RANK_RANGES=1), the CLI's multi-rank tests under it
cd "$GRAFT_REPO_ROOT"
export FLX_CLI_RANK_RANGES=1
timeout 60 python -m pytest tests/test_gpu_cli.py -q -m gpu -x -k "sinks_with_forked_ranks or verbose_matches_reference_stderr" 2>&1 | tail -4 | cut -c1-400
