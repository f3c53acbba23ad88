cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -c "import torch; print(torch.cuda.is_available())"
timeout 1500 python -m pytest tests/test_gpu_coop_cg.py tests/test_gpu_reorder.py -q -m gpu --timeout=900 -x > gpurun_out/r04/t_coop.log 2>&1; echo rc=$?
tail -4 gpurun_out/r04/t_coop.log
for g in -40 -64 -80 -90; do for e in 1 0; do echo "== grid $g coop=$e"; SGM_CG_COOP=$e timeout 300 python tools/ildu_bench.py $g cg 2>&1 | grep '^{' | cut -c1-200; done; done
