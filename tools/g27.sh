cd $GRAFT_REPO_ROOT
python -c "import torch; print(torch.cuda.is_available())"
for e in "X=1" "SGM_CG_COOP=0" "SGM_CG_COOP_XCD=0"; do
  echo "== $e"; env $e timeout 300 python tools/probes/bicg_check.py 2>&1 | grep -v amdgpu.ids
done
timeout 1500 python -m pytest tests/test_gpu_coop_cg.py -q -m gpu --timeout=900 -x -k "bicgstab" 2>&1 | tail -15
