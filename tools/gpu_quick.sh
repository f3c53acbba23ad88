cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ildu" > gpurun_out/r02/quick.log 2>&1; echo quick=$?
tail -3 gpurun_out/r02/quick.log
for bh in 64 128 256; do echo "bh $bh"; SGM_TRSV_GRID_BH=$bh python tools/ildu_bench.py 1000 ildu0 2>&1 | tail -1; done
SGM_TRSV_GRID_BH=128 python tools/ildu_bench.py 2000 ildu0 2>&1 | tail -1
SGM_TRSV_GRID_BH=128 python tools/ildu_bench.py 500 ildu0 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "2-laplace3d or 3-random" 2>&1 | tail -3
