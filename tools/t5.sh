cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -k "ildu or precond or golden" 2>&1 | grep -E "passed|failed|rror" | head -5
python tools/ildu_bench.py 1000 | tail -1
