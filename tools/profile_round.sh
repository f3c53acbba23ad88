# Round profile: kernel stats + HBM counters of the bench command (run on the GPU box).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_round
rm -rf $OUT; mkdir -p $OUT
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err; echo bench=$?
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 50 --warmup 5 --cg-steps 30 --no-cpu > $OUT/stats.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python bench.py --steps 5 --warmup 1 --cg-steps 0 --no-cpu > $OUT/pmc1.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python bench.py --steps 5 --warmup 1 --cg-steps 0 --no-cpu > $OUT/pmc2.log 2>&1
# ILDU(0)-PCG on the 1000^2 grid: per-kernel times of the level walkers
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_ildu -- python tools/ildu_bench.py 1000 ildu0 > $OUT/stats_ildu.log 2>&1
cat $OUT/bench.json
# secondary configs of BASELINE.json (C3 BiCGStab/GMRES, C4 ELLPACK, C5 464^3 on one GPU): per-kernel times
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_configs -- python tools/bench_configs.py --configs c3,c4,c5 > $OUT/configs.log 2>&1
grep '^{' $OUT/configs.log > $OUT/configs.jsonl
