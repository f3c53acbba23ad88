cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "overlap or partitioned or rccl" 2>&1 | grep -E "passed|failed|Error"
python bench.py --workload c5 --c5-edge 200 --steps 20 --warmup 5 --cg-steps 20 --no-cpu 2>&1 | tail -1 | cut -c1-600
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --force-dist --workload c5 --c5-edge 200 --steps 20 --warmup 5 --no-cpu --cg-steps 20 2>&1 | tail -1 | cut -c1-600
python bench.py --steps 50 --warmup 5 --cg-steps 50 2>&1 | tail -1 | cut -c1-300
