#!/bin/bash
# Runs on the GPU box (via gpurun): benches + rocprofv3 passes; leaves everything under gpurun_out/r01/.
# Copy the summaries into profiles/ afterwards with tools/summarise_profiles.py.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r01
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python $R/bench.py --config c2 --steps 200 > $OUT/bench_c2.json 2>> $OUT/bench.err
python $R/bench.py --config c3 --steps 100 > $OUT/bench_c3.json 2>> $OUT/bench.err
python $R/bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_c5.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/stats_c4 -o c4 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_c4_profiled.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $OUT/stats_c3 -o c3 -- python $R/bench.py --config c3 --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/stats_c2 -o c2 -- python $R/bench.py --config c2 --steps 50 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o c4 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o c4 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_c3 -o c3 -- python $R/bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_c3 -o c3 -- python $R/bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_sq1 -o c4 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o c4 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
for l in warp-rnnt warp-rnnt-gather warp-rnnt-compact warp-rnnt-fused; do
  timeout 300 python $R/tools/benchmark_table.py --loss $l --markdown $OUT/table_$l.md > $OUT/table_$l.log 2>&1
done
# cache / memory counters of the loss kernels (gather, lattice, gradients), one --pmc pass per counter group
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum GRBM_GUI_ACTIVE" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_gather$i -o g -- python $R/tools/gather_probe.py $R/warp_rnnt_amd/libwarp_rnnt_amd.so > /dev/null 2>&1
done
ls $OUT
