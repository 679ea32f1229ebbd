#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05c6
mkdir -p $OUT
cd $R
export ROUTES_NO_PD=1
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/log.txt
timeout 300 python tools/lattice_routes.py > $OUT/lattice_routes.txt 2>&1
echo "routes rc=$?" >> $OUT/log.txt
RNNT_WD_K16_FROM_T=100000 timeout 100 python tools/lattice_routes.py 16,1500,64 16,1500,300 16,1500,512 8,3000,500 > $OUT/lattice_routes_k8.txt 2>&1
timeout 120 python tools/wd_trace.py 16 1500 300 2>&1 | grep -v amdgpu > $OUT/wd_trace_c4.txt
echo "trace rc=$?" >> $OUT/log.txt
timeout 300 python tools/cabi_probe.py c2 c4 2>&1 | grep -v amdgpu > $OUT/cabi_probe.txt
echo "cabi rc=$?" >> $OUT/log.txt
timeout 300 python bench.py --no-cpu-baseline --steps 30 > $OUT/bench_c4.json 2> $OUT/bench_c4.err
echo "bench rc=$?" >> $OUT/log.txt
timeout 200 python bench.py --config c2 --steps 300 --warmup 20 --no-cpu-baseline > $OUT/bench_c2.json 2>> $OUT/bench_c4.err
echo "bench c2 rc=$?" >> $OUT/log.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
echo "smoke rc=$?" >> $OUT/log.txt
tail -4 $OUT/pytest_gpu.txt | cut -c1-200; cat $OUT/log.txt; grep -v amdgpu $OUT/lattice_routes.txt | cut -c1-190; grep -v amdgpu $OUT/lattice_routes_k8.txt | cut -c1-150; cat $OUT/cabi_probe.txt; tail -3 $OUT/smoke.txt
