cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_v4
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q --timeout=900 -k "gmres or scattered or composite or lanczos or ell_column" > gpurun_out/r05_v4/t0.log 2>&1; echo t0=$?; tail -15 gpurun_out/r05_v4/t0.log
for f in "" "--gmres-cgs2" "--gmres-mgs"; do timeout 600 python tools/bench_configs.py --configs c3 $f 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['gmres30'], d['bicgstab']['iters_per_s'])"; done
timeout 600 python tools/probes/scattered_csr.py 2>&1 | grep '^{'
timeout 900 python tools/probes/ildu_parts.py 3162 8 300 2>&1 | grep '^{'
timeout 900 python -m pytest tests/test_gpu_multirank.py -x -q --timeout=900 -k "share_one_gpu" > gpurun_out/r05_v4/t1.log 2>&1; echo t1=$?; tail -8 gpurun_out/r05_v4/t1.log
