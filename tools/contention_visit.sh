# The single-process GPU suites while two other processes keep the GPU busy with matrix products: a stream-ordering race that a
# quiet GPU hides (a blocking null-stream copy right after a kernel on the library's non-blocking stream, ...) shows as a wrong
# result here.  One GPU-box visit; round 6 found the breadth-first copy-back this way.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
cat > /tmp/busy.py <<'P'
import torch, time, sys
a = torch.randn(4096, 4096, device='cuda'); t0 = time.time()
while time.time() - t0 < float(sys.argv[1]):
    for _ in range(20): b = a @ a
    torch.cuda.synchronize()
P
python /tmp/busy.py ${BUSY_S:-420} & B1=$!
python /tmp/busy.py ${BUSY_S:-420} & B2=$!
sleep 5
timeout ${PT:-400} python -m pytest tests/test_gpu_parity.py tests/test_gpu_reorder.py tests/test_gpu_boundary.py tests/test_gpu_coop_cg.py -n 4 -q --tb=line --timeout=300 -k "${KSEL:-not full_size and not c2_size}" > gpurun_out/r06/contention.log 2>&1; echo suite=$?
kill $B1 $B2 2>/dev/null; wait $B1 $B2 2>/dev/null
grep -v "dist-packages" gpurun_out/r06/contention.log | grep "^/\|^FAILED\|passed\|failed" | cut -c1-240 | head -40
