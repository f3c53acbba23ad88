cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 300 sigma_amd/fortran/surface_test_hip > gpurun_out/r04/surface_test_hip.log 2>&1; echo surface=$?
tail -12 gpurun_out/r04/surface_test_hip.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fortran or reference_side" > gpurun_out/r04/t_fortran.log 2>&1; echo fortran=$?
tail -30 gpurun_out/r04/t_fortran.log
