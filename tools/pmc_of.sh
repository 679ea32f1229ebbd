#!/bin/bash
# SQ counters of one command's kernels, two passes (never combined with other trace domains):
#   bash tools/pmc_of.sh <tag> <kernel substring> -- <command ...>      -> gpurun_out/pmc_<tag>.txt
tag=$1; pat=$2; shift 3
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/pmc_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/p1 -o p -- "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/p2 -o p -- "$@" > /dev/null 2>&1
python3 - "$OUT" "$pat" > $R/gpurun_out/pmc_$tag.txt <<'PY'
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print(f"{k:60s} {c:24s} dispatches {n:4d}  mean {v / n:16.1f}")
PY
rm -rf $OUT
cat $R/gpurun_out/pmc_$tag.txt
