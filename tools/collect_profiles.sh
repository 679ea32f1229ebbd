#!/bin/bash
# Runs on the GPU box (via gpurun): benches + rocprofv3 passes; leaves everything under gpurun_out/$TAG/.
# Copy the summaries into profiles/ afterwards with `python tools/summarise_profiles.py $TAG`.
#   --pmc passes are separate from each other and never combined with other trace domains.
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python $R/bench.py --config c2 --steps 500 --warmup 20 > $OUT/bench_c2.json 2>> $OUT/bench.err
python $R/bench.py --config c3 --steps 100 > $OUT/bench_c3.json 2>> $OUT/bench.err
python $R/bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_c5.json 2>> $OUT/bench.err
RNNT_WD_K16_FROM_T=1000000 python $R/bench.py --no-cpu-baseline > $OUT/bench_c4_blocks_of_8.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/stats_c4 -o c4 -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_c4_profiled.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $OUT/stats_c3 -o c3 -- python $R/bench.py --config c3 --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/stats_c2 -o c2 -- python $R/bench.py --config c2 --steps 100 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
# SLIM=1: only the bench lines and the kernel-trace summaries (the counter passes of an earlier collection under the same
# tag stay where they are; tools/summarise_profiles.py reads whatever gpurun_out/$TAG holds)
if [ "${SLIM:-0}" = 1 ]; then
  cd $R
  python $R/bench.py --no-cpu-baseline --rccl-group > $OUT/bench_c4_rccl_group.json 2>> $OUT/bench.err
  python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --preload-ms 0 > $OUT/bench_c4_cold_start.json 2>> $OUT/bench.err
  ls $OUT
  exit 0
fi
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o c4 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o c4 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_c3 -o c3 -- python $R/bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_c3 -o c3 -- python $R/bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_sq1 -o c4 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o c4 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
# gather path: TA / TCP / TCC counters of the loss entry's kernels (tools/gather_probe.py = rnnt_amd_loss on dense log-probs)
i=0
for set in "TA_BUSY_avr TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum" "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUSY_avr" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_gather$i -o g -- python $R/tools/gather_probe.py $R/warp_rnnt_amd/libwarp_rnnt_amd.so > /dev/null 2>&1
done
# what FETCH_SIZE tallies per touched 128-byte line (one dword per 64 / 128 / 256 / 512 bytes over 1.44 GB)
PROBES_ONLY=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_probe -o p -- $R/tools/ubench/gather_variants > /dev/null 2>&1
$R/tools/ubench/gather_variants > $OUT/ubench_gather_variants.txt 2>&1
cd $R
# parity at BASELINE sizes: the shipped build, the same with the log-domain lattice kernels only, and the libm build
rm -f $OUT/parity_errors.json
# (the test file runs every configuration on both lattice routes itself since round 3: rows carry "route")
RNNT_PARITY_TABLE=$OUT/parity_errors.json RNNT_PARITY_BUILD="shipped" python -m pytest tests/test_gpu_baseline_sizes.py -q > $OUT/parity_default.log 2>&1
# (the libm build of the log-domain lattice -- the reference's own log1pf(expf()) -- is in profiles/r02_parity_errors.json:
#  within 5e-4 of the shipped log-domain route at every size)
# lattice kernels alone: probability domain, log domain on one workgroup per sweep / per column block
python tools/lattice_routes.py > $OUT/lattice_routes.txt 2>&1
# the yardsticks: copy and read-only streams of the part; the in-place log-softmax against an in-place copy at c5's V from 2 GB
# to 130 GB; the N=128 single-GPU anchor of the strong-scaling line
tools/ubench/copy_rate > $OUT/ubench_copy_rate.txt 2>&1
LSM_COPY=1 LSM_INPLACE=1 python tools/lsm_rate.py 10000:1.92 10000:16 10000:64 10000:130 2>&1 | grep -v amdgpu > $OUT/lsm_rate_by_size.txt
python $R/bench.py --config c4 --global-batch 128 --no-cpu-baseline --steps 20 > $OUT/bench_c4_n128.json 2>> $OUT/bench.err
# the reference-named C entry points the way the reference's binding calls them
python tools/cabi_probe.py c2 c4 2>&1 | grep -v amdgpu > $OUT/cabi_probe.txt
# interval-by-interval timeline of the distributed log-domain kernel (diagnostics build, if it travelled)
if [ -f tools/_probe/wdstats/lib.so ]; then
  python tools/wd_trace.py 16 1500 300 2>&1 | grep -v amdgpu > $OUT/wd_trace_c4.txt
fi
(python tools/graph_probe.py 16 150 40 28 0 1000; python tools/graph_probe.py 16 1500 300 50 1) 2>&1 | grep "N=" > $OUT/graph_probe.txt
python tools/host_overhead.py 2>&1 | grep -v amdgpu > $OUT/host_overhead.txt
(echo "== ctypes fallback"; WARP_RNNT_AMD_NO_NATIVE_BINDING=1 python tools/host_overhead.py 2>&1 | grep -v amdgpu) >> $OUT/host_overhead.txt
python tools/compact_host_probe.py 2>&1 | grep -v amdgpu > $OUT/compact_host_probe.txt
python $R/bench.py --no-cpu-baseline --rccl-group > $OUT/bench_c4_rccl_group.json 2>> $OUT/bench.err
for l in warp-rnnt warp-rnnt-gather warp-rnnt-fused warp-rnnt-compact; do
  timeout 200 python $R/tools/benchmark_table.py --loss $l --markdown $OUT/table_$l.md > $OUT/table_$l.log 2>&1
done
ls $OUT
# shape map off BASELINE's five points (one kernel-trace pass)
mkdir -p $OUT/shape_map
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/shape_map -o map -- python $R/tools/shape_map.py run > $OUT/shape_map/manifest.json 2> $OUT/shape_map/err.txt)
python tools/shape_map.py report $OUT/shape_map/map_kernel_trace.csv $OUT/shape_map/manifest.json > $OUT/shape_map.md 2>> $OUT/bench.err
rm -f $OUT/shape_map/map_kernel_trace.csv
ls $OUT
