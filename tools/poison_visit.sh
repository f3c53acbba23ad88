# One GPU-box visit with the library rebuilt IN PLACE ON THE BOX with -DSGM_POISON_ALLOC (every device allocation starts as NaN /
# -1, synchronously): a read of memory nothing has written shows deterministically.  Nothing of this build travels back.
#   KSEL="<pytest -k expression>" MULTI="<-k expression for tests/test_gpu_multirank.py>" bash tools/poison_visit.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06/poison
make -C sigma_amd/csrc clean > /dev/null; make -C sigma_amd/csrc POISON=1 -j32 > gpurun_out/r06/poison/build.log 2>&1; echo build=$?
timeout ${PT:-600} python -m pytest tests/test_gpu_parity.py tests/test_gpu_reorder.py tests/test_gpu_boundary.py tests/test_gpu_coop_cg.py -n 4 -q --tb=line --timeout=300 -k "${KSEL:-not full_size}" > gpurun_out/r06/poison/parity.log 2>&1; echo parity=$?
grep -v "dist-packages" gpurun_out/r06/poison/parity.log | grep "^/\|^FAILED\|passed\|failed" | cut -c1-260 | head -60
if [ -n "$MULTI" ]; then
timeout ${MT:-600} python -m pytest tests/test_gpu_multirank.py -q --tb=short --timeout=300 -k "$MULTI" > gpurun_out/r06/poison/multirank.log 2>&1; echo multirank=$?
grep "Error\|assert\|passed\|failed" gpurun_out/r06/poison/multirank.log | cut -c1-260 | head -40
fi
