# PMC passes over one command, one counter group per pass; prints per-kernel means for kernels matching $KERNEL.
#   KERNEL=k_csr_rl OUT=gpurun_out/pmc_rl bash tools/pmc_kernel.sh python tools/general_rows.py
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=${OUT:-gpurun_out/pmc_kernel}; rm -rf $OUT; mkdir -p $OUT
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/g$i -- "$@" > $OUT/g$i.log 2>&1
  echo "group $i ($C) rc=$?"
done
python - <<PY
import csv, glob, collections, os
kern = os.environ.get("KERNEL", "k_")
for f in sorted(glob.glob("$OUT/g*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][:50], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(k, c, "n=%d mean=%.5g" % (len(v), sum(v) / len(v)))
PY
