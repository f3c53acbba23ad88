# Round profile (run on the GPU box): the bench line, kernel stats + HBM counters of the bench command,
# and the per-kernel times of the other BASELINE configs.  tools/collect_profiles.py condenses it into profiles/<tag>/.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_round
rm -rf $OUT; mkdir -p $OUT
T0=$(date +%s); timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo bench=$? wall=$(( $(date +%s) - T0 ))s | tee $OUT/bench_wall.txt
BARGS="--steps 2 --warmup 1 --spmv-per-step 64 --cg-steps 30 --no-cpu --no-c5 --no-c3 --no-c4 --no-dist-overhead --no-pcg --no-ceilings"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $BARGS > $OUT/stats.log 2>&1
PARGS="--steps 1 --warmup 1 --spmv-per-step 4 --cg-steps 0 --no-cpu --no-c5 --no-c3 --no-c4 --no-variants --no-dist-overhead --no-pcg --no-ceilings"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $PARGS > $OUT/pmc1.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $PARGS > $OUT/pmc2.log 2>&1
# HBM bytes of the CG update kernels (k_elem<FCgR / FCgPX>) and of the C5 product (k_csr_sl<7>, 464^3): their own PMC passes
QARGS="--steps 1 --warmup 1 --spmv-per-step 2 --cg-steps 10 --c5-cg-steps 4 --no-cpu --no-c3 --no-c4 --no-variants --no-dist-overhead --no-pcg --no-ceilings --no-c5-parts"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_cg_c5 -- python3 bench.py $QARGS > $OUT/pmc3.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_cg_c5 -- python3 bench.py $QARGS > $OUT/pmc4.log 2>&1
# ILDU(0)-PCG on the 1000^2 grid: per-kernel times of the triangular solves
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_ildu -- python3 tools/ildu_bench.py 1000 ildu0 > $OUT/stats_ildu.log 2>&1
# ILDU(0)-PCG on the 100^3 grid: the slab-pipelined triangular solves
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_ildu3 -- python3 tools/ildu_bench.py -100 ildu0 > $OUT/stats_ildu3.log 2>&1 < /dev/null
# colour-ordered ILDU(0)-PCG at C2 size (row-space level sweeps): per-kernel times
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_ildu_colour -- python3 tools/ildu_bench.py 3162 ildu0 colour > $OUT/stats_ildu_colour.log 2>&1 < /dev/null
# HBM bytes of the colour-ordered sweeps (k_trsv_rows_cg2, k_trsv_rows2) and the product on the permuted matrix
bash tools/probes/pmc_colour_ildu.sh > $OUT/pmc_colour_ildu.txt 2>&1
# CG per iteration on mid-sized grids (one-workgroup / one-XCD / all-CU cooperative kernels, launch loop beyond), C1, and
# where an iteration of the cooperative kernel spends its time (phase timers: the -DSGM_COOP_PROBE build of the library)
NXS=32,64,100,181,256,300,316,362,500,700,1000,1100,1448,1500,2000 SOLVERS=cg KRYLOV_GRAPH=1 timeout 600 python tools/cg_small.py 2>&1 | grep '^{' > $OUT/cg_small_coop.jsonl
NXS=100,316,1000,1448,2000 SOLVERS=cg KRYLOV_GRAPH=1 NO_C1=1 LAUNCH_LOOP=1 timeout 600 python tools/cg_small.py 2>&1 | grep '^{' > $OUT/cg_small_launch_loop.jsonl
NXS=32,100,316,1000 SOLVERS=bicgstab KRYLOV_GRAPH=1 NO_C1=1 timeout 600 python tools/cg_small.py 2>&1 | grep '^{' > $OUT/bicgstab_small.jsonl
if [ -f tools/probes/libsigma_hip_probe.so ]; then
  NXS=100,256,300,500,1000 timeout 300 python tools/probes/coop_probe.py 2>&1 | grep '^{' > $OUT/coop_probe.jsonl
fi
[ -x tools/probes/wave_sum_probe ] && ./tools/probes/wave_sum_probe > $OUT/wave_sum_probe.txt 2>&1
# time to solution: CG / Jacobi-PCG / ILDU(0)-PCG in natural and colour order, with the setup phases (SGM_PC_TIMING)
for a in "1000 cg,jacobi,ildu0,ildu0_reorder" "1000 cg,ildu0 colour" "3162 cg,ildu0,ildu0_reorder" "3162 cg,ildu0 colour" "-100 cg,jacobi,ildu0,ildu0_reorder" "-100 cg,ildu0 colour"; do
  echo "== tools/ildu_bench.py $a"
  SGM_PC_TIMING=1 timeout 600 python tools/ildu_bench.py $a 2>&1 | grep -E '^\{|ildu setup'
done > $OUT/time_to_solution.log 2>&1
# C3 GMRES(30): low-synchronisation CGS-2 (default) vs modified Gram-Schmidt (launch counts per step come out of the Calls column)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3_lowsync -- python3 tools/bench_configs.py --configs c3 > $OUT/c3_lowsync.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3_mgs -- python3 tools/bench_configs.py --configs c3 --gmres-mgs > $OUT/c3_mgs.log 2>&1
# a CSR matrix with scattered columns (n = 5e6, 8..32 per row): the column-blocked form against the row kernels
timeout 600 python tools/probes/scattered_csr.py 2>&1 | grep '^{' > $OUT/scattered_csr.json
# colour-ordered ILDU(0)-PCG at C2 size on 8 in-process parts against one part: wall clock and kernel-time sums
bash tools/probes/ildu_parts_stats.sh > /dev/null 2>&1
cp gpurun_out/r05_parts/summary.txt $OUT/ildu_colour_parts_kernel_sums.txt
# the streaming ceilings and the counter passes of the C5 product (tools/probes/ceilings_r05.sh)
bash tools/probes/ceilings_r05.sh > /dev/null 2>&1
cp gpurun_out/r05_ceilings/stream_ceilings.txt gpurun_out/r05_ceilings/c5_counters.txt $OUT/
# short campaigns of the fuzzers on this tree (the solver fuzzer holds every seventh one-part system in ELLPACK since round 6)
{ echo "== tests/fuzz_solvers.py 240 s from seed 600000"; timeout 400 python tests/fuzz_solvers.py 240 600000 2>&1 | tail -4;
  echo "== tests/fuzz_formats.py 120 s from seed 610000"; timeout 300 python tests/fuzz_formats.py 120 610000 2>&1 | tail -4; } > $OUT/fuzz_campaigns.txt 2>&1
# C4 / C5: per-kernel times
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_configs -- python3 tools/bench_configs.py --configs c4,c5 > $OUT/configs.log 2>&1
grep -h '^{' $OUT/c3_lowsync.log $OUT/c3_mgs.log $OUT/configs.log > $OUT/configs.jsonl
cat $OUT/bench.json | cut -c1-600
# the bench line once more, now that this round's PMC passes exist: condense them on the box (profiles/r03/pmc_hbm_traffic.json
# with the fingerprint of the sources that just ran) so that `roofline.traffic` of the line is this run's own figure
python tools/collect_profiles.py ${TAG:-r06} > /dev/null 2>&1
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo bench_with_traffic=$?
# what travels back is capped at 64 MiB: the per-dispatch traces are not needed once the stats exist
find $OUT -name "*kernel_trace.csv" -delete
du -sh $OUT
