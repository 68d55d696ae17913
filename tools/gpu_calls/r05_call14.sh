cd "$GRAFT_REPO_ROOT"
bash tools/r05_final.sh 2>&1 | tail -40
