cd $GRAFT_REPO_ROOT
python -c "import torch; print(torch.cuda.is_available())"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_coop_cg.py -q -m gpu --timeout=900 -x -k "small or bicg or advection or golden or solver" 2>&1 | tail -4
NXS=32,64 SOLVERS=bicgstab KRYLOV_GRAPH=1 NO_C1=1 timeout 300 python tools/cg_small.py 2>&1 | grep '^{'
