cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -28
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
