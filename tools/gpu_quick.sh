cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ildu or precond or solvers_golden or colour" > gpurun_out/r02/quick.log 2>&1; echo quick=$?
tail -8 gpurun_out/r02/quick.log
for g in 1 0; do SGM_ILDU_GRID=$g python tools/ildu_bench.py 1000 ildu0 2>&1 | tail -2; done
SGM_ILDU_GRID=1 python tools/ildu_bench.py 2000 ildu0 2>&1 | tail -1
SGM_ILDU_GRID=1 python tools/ildu_bench.py 500 ildu0 2>&1 | tail -1
for bh in 128 512; do SGM_TRSV_GRID_BH=$bh python tools/ildu_bench.py 1000 ildu0 2>&1 | tail -1; done
for m in 5 6 7; do SGM_SPMV_CFG=256,2,1,0,$m,0 python bench.py --steps 5 --warmup 2 --no-cpu --no-c5 --no-variants --cg-steps 100 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('remap', $m, o['roofline']['avg_launch_ms'], o['roofline']['cold_launch_ms'], o['cg']['iters_per_s'])"; done
