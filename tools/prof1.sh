cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export SGM_BENCH_CG=100
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cg -- tools/spmv_bench 3162 3162 20 > gpurun_out/prof_cg.log 2>&1
cat gpurun_out/prof_cg/*/*kernel_stats.csv | cut -c1-60,200- | head -12
cat gpurun_out/prof_cg/*/*kernel_stats.csv | awk -F'","' '{print substr($1,1,70), $2, $4}' | head -12
