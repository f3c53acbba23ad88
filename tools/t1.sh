cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error" 
./tools/spmv_bench 3162 3162 50
