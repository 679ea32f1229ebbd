#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05c5
mkdir -p $OUT
cd $R
export ROUTES_NO_PD=1
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/log.txt
timeout 300 python tools/lattice_routes.py > $OUT/lattice_routes.txt 2>&1
echo "routes rc=$?" >> $OUT/log.txt
WARP_RNNT_AMD_LIB=$R/warp_rnnt_amd/libwarp_rnnt_amd_wd_k16.so timeout 200 python tools/lattice_routes.py 16,1500,64 16,1500,300 16,1500,512 32,1500,300 16,700,100 16,150,40 > $OUT/lattice_routes_k16.txt 2>&1
echo "routes k16 rc=$?" >> $OUT/log.txt
timeout 120 python tools/wd_trace.py 16 1500 300 2>&1 | grep -v amdgpu > $OUT/wd_trace_c4.txt
echo "trace rc=$?" >> $OUT/log.txt
timeout 300 python bench.py --no-cpu-baseline --steps 30 > $OUT/bench_c4.json 2> $OUT/bench_c4.err
echo "bench rc=$?" >> $OUT/log.txt
timeout 200 python bench.py --config c2 --steps 300 --warmup 20 --no-cpu-baseline > $OUT/bench_c2.json 2>> $OUT/bench_c4.err
echo "bench c2 rc=$?" >> $OUT/log.txt
tail -5 $OUT/pytest_gpu.txt; cat $OUT/log.txt; grep -v amdgpu $OUT/lattice_routes.txt | cut -c1-190; grep -v amdgpu $OUT/lattice_routes_k16.txt | cut -c1-150
