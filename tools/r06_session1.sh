cd $GRAFT_REPO_ROOT
bash tools/r06_overlap.sh 2>&1 | tail -30
