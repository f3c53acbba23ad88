cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -c "import torch; print(torch.cuda.is_available())"
timeout 900 python -m pytest tests/test_gpu_reorder.py -q -m gpu --timeout=600 -x -s > gpurun_out/r04/t_reorder.log 2>&1; echo rc=$?
grep -n "greedy_color_ordering at\|passed\|failed\|Error" gpurun_out/r04/t_reorder.log | head
tail -5 gpurun_out/r04/t_reorder.log
echo "== coop probe (one XCD where it fits)"; timeout 300 python tools/probes/coop_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/coop_probe_xl.jsonl
echo "== coop probe (all CUs)"; SGM_CG_COOP_XCD=0 timeout 300 python tools/probes/coop_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/coop_probe_all.jsonl
