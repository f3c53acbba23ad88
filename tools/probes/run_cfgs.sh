cd $GRAFT_REPO_ROOT
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-cpu --cg-steps 30 > gpurun_out/bench_forcedist.json 2> gpurun_out/bench_forcedist.err; echo forcedist=$?; tail -3 gpurun_out/bench_forcedist.err; cut -c1-400 gpurun_out/bench_forcedist.json
timeout 1200 python tools/bench_configs.py --configs c3,c4,c5 > gpurun_out/configs.jsonl 2> gpurun_out/configs.err; echo cfg=$?; tail -5 gpurun_out/configs.err; cat gpurun_out/configs.jsonl
