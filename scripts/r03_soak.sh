cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 900 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror" | tail -2; done
timeout 600 python3 scripts/soak_search.py 2>&1 | tail -3
timeout 600 python3 scripts/soak_update.py 2>&1 | tail -3
