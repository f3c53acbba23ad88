cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for f in test_gpu_boundary test_gpu_reorder test_gpu_coop_cg; do
  timeout 900 python -m pytest tests/$f.py -q -m gpu --durations=8 --timeout=240 > gpurun_out/r04/t_$f.log 2>&1; echo $f=$?
  grep -E "passed|failed|Timeout|slowest|^[0-9.]+s (call|setup)" gpurun_out/r04/t_$f.log | head -14
  grep -E "^(E  |>  )" gpurun_out/r04/t_$f.log | head -12
done
