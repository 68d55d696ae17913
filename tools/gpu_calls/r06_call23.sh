cd "$GRAFT_REPO_ROOT"; bash tools/r06_final.sh 2>&1 | tail -40
