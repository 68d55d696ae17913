cd "$GRAFT_REPO_ROOT"; bash tools/r06_final.sh skip-tests 2>&1 | tail -30
