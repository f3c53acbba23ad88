cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error|assert" | head
