cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo pytest=$?; tail -5 gpurun_out/pytest_gpu.log
python bench.py --steps 100 --warmup 10 > gpurun_out/bench2.json 2> gpurun_out/bench2.err; echo bench=$?; cat gpurun_out/bench2.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -- python bench.py --steps 50 --warmup 5 --cg-steps 20 --no-cpu > gpurun_out/prof_stdout.log 2>&1
find gpurun_out/prof_r01 -name "*stats*" | head; 
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 5 --warmup 1 --cg-steps 0 --no-cpu > gpurun_out/pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python bench.py --steps 5 --warmup 1 --cg-steps 0 --no-cpu > gpurun_out/pmc2.log 2>&1
ls -R gpurun_out | head -40
