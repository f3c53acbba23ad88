# C4 quick check on the GPU box: column-blocked ELLPACK tests + the band A/B + per-kernel times
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ell_column_blocked or full_size_c4" 2>&1 | tail -3
C4_SETTINGS="-1,512,0" bash tools/prof_c4_band.sh 2>&1 | tail -6
