cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -c "import torch; print(torch.cuda.is_available())"
timeout 1500 python -m pytest tests/test_gpu_reorder.py tests/test_gpu_parity.py -q -m gpu --timeout=900 -x -k "reorder or colour or ildu or ldu or rows or graph" > gpurun_out/r04/t_rows.log 2>&1; echo rc=$?
tail -4 gpurun_out/r04/t_rows.log
for a in "1000 cg,ildu0 colour" "1000 cg,ildu0_reorder" "316 cg,ildu0_reorder" "3162 ildu0_reorder"; do echo "== $a"; timeout 600 python tools/ildu_bench.py $a 2>&1 | grep '^{'; done
echo "== no graphs"; for a in "1000 ildu0 colour"; do SGM_KRYLOV_GRAPH=0 timeout 600 python tools/ildu_bench.py $a 2>&1 | grep '^{'; done
