cd $GRAFT_REPO_ROOT && timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "full_size" 2>&1 | grep "^E\|assert" | cut -c1-220 | head -12
