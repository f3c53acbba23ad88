cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -k "ildu or precond or golden or Fortran or fortran" 2>&1 | grep -E "passed|failed|rror" | head -5
python tools/ildu_bench.py 1000 | tail -1 | cut -c1-200
python tools/ildu_bench.py 3162 | tail -2 | cut -c1-200
