cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -c "import torch; print(torch.cuda.is_available())"
timeout 2400 python -m pytest tests -q -m gpu --timeout=900 --durations=8 > gpurun_out/r04/gpu_tests.log 2>&1; echo gputests=$?
tail -20 gpurun_out/r04/gpu_tests.log
export NXS=300,316,330,362 SOLVERS=cg KRYLOV_GRAPH=1 NO_C1=1
timeout 300 python tools/cg_small.py 2>&1 | grep -v amdgpu.ids
