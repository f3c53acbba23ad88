cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gmres" > gpurun_out/r02/quick.log 2>&1; echo quick=$?
tail -25 gpurun_out/r02/quick.log
timeout 600 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "3-random or 2-laplace3d" > gpurun_out/r02/quick2.log 2>&1; echo quick2=$?
tail -5 gpurun_out/r02/quick2.log
for o in 1 0; do SGM_GMRES_CGS2=$o timeout 600 python tools/bench_configs.py --configs c3 2>&1 | grep '^{' ; done
