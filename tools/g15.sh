cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -c "import torch; print(torch.cuda.is_available())"
timeout 1500 python -m pytest tests/test_gpu_reorder.py tests/test_gpu_parity.py -q -m gpu --timeout=900 -x -k "reorder or colour or ildu or ldu or rows" > gpurun_out/r04/t_rows.log 2>&1; echo rc=$?
tail -6 gpurun_out/r04/t_rows.log
bash tools/probes/prof_colour_ildu.sh 3162
