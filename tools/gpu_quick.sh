cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ildu or precond or colour or solvers_golden" > gpurun_out/r02/quick.log 2>&1; echo quick=$?
tail -3 gpurun_out/r02/quick.log
for g in 250 500 1000 1400 2000; do timeout 300 python tools/ildu_bench.py $g ildu0 2>&1 | tail -1 | cut -c1-140; done
