# Round-5 visit 1: the streaming ceilings this tree is graded against (0.6 / 2 / 7.6 GiB footprints), and what binds the C5
# product (k_csr_sl<7>, 464^3, 7.6 GiB) -- address-translation and L2 counters of it next to the same counters of a plain
# streamed read of the same footprint.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_ceilings; rm -rf $OUT; mkdir -p $OUT
for mib in 600 2048 7600; do echo "## footprint $mib MiB (read buffer)"; ./tools/stream_bench $mib; done > $OUT/stream_ceilings.txt 2>&1
tail -14 $OUT/stream_ceilings.txt
sum() { python3 - "$1" "$2" <<'PY'
import csv, glob, collections, sys
pat = sys.argv[2]
for f in sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0].replace("void sgm::", "")[:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print("  ", k, c, "launches=%d per_launch=%.6g" % (len(v), sum(v) / len(v)))
PY
}
i=0
for C in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
         "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  echo "== pass $i: $C"
  timeout 400 rocprofv3 --pmc $C --output-format csv -d $OUT/c5_p$i -- python3 tools/probes/c5_product.py 464 4 > $OUT/c5_p$i.log 2>&1
  echo "-- C5 product"; sum $OUT/c5_p$i k_csr_sl
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $OUT/st_p$i -- ./tools/stream_bench 7600 > $OUT/st_p$i.log 2>&1
  echo "-- plain streams, 7600 MiB"; sum $OUT/st_p$i k_read; sum $OUT/st_p$i k_mix
done > $OUT/c5_counters.txt 2>&1
find $OUT -name "*counter_collection.csv" -size +2M -delete
find $OUT -name "*.db" -delete
cat $OUT/c5_counters.txt | cut -c1-200 | tail -80
du -sh $OUT
