#!/bin/bash
timeout 1200 python -m pytest tests -q -m gpu -x -k "${1:-pc_info or reorder or fuzzer}" --timeout=600 2>&1 | tail -4
