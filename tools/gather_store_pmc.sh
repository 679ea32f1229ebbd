#!/bin/bash
# Counter passes over tools/ubench/gather_r06 in its GATHER_R06_PMC mode (three kernels: the shipped gather, the same with its
# stores kept in L2, the tile-major form): what the memory side does differently when the 57.6 MB of stores go to DRAM.
# Separate --pmc passes, never combined with other trace domains.  Summary -> $OUT/gather_store_pmc.csv
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r06}
mkdir -p $OUT/pmc_gstore
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_CYCLE_sum" \
           "TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_WRITE_sum" \
           "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  GATHER_R06_PMC=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_gstore/p$i -o g -- $R/tools/ubench/gather_r06 > /dev/null 2>&1
done
python - <<PY
import collections, csv, glob
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_gstore/p*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_tile" not in k:
            continue
        tag = "tile-major" if "-1>" in k.replace(" ", "") else ("hot" if "true,false,0>" in k.replace(" ", "") or ", true, false" in k else "shipped")
        agg[(tag, r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("$OUT/gather_store_pmc.csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["variant", "counter", "dispatches", "mean_per_dispatch"])
    for (t, c), v in sorted(agg.items()):
        w.writerow([t, c, len(v), round(sum(v) / len(v), 1)])
print(open("$OUT/gather_store_pmc.csv").read())
PY
# the kernel names, to check the tagging above
grep -h -o "k_tile<[^>]*>" $OUT/pmc_gstore/p1/*_counter_collection.csv | sort | uniq -c
rm -rf $OUT/pmc_gstore
