#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r05
cd $R
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 --timeout 600 -p no:cacheprovider > $R/gpurun_out/r05/pytest_gpu.txt 2>&1
echo "pytest rc=$?" > $R/gpurun_out/r05/log.txt
python -c "import __graft_entry__ as g; g.smoke()" > $R/gpurun_out/r05/smoke.txt 2>&1
echo "smoke rc=$?" >> $R/gpurun_out/r05/log.txt
bash tools/collect_profiles.sh r05 > $R/gpurun_out/r05/collect.log 2>&1
echo "collect rc=$?" >> $R/gpurun_out/r05/log.txt
# the raw traces exceed what gpurun copies back: summarise here, keep the summaries and the small files
PROFILES_DST=$R/gpurun_out/r05/summary python tools/summarise_profiles.py r05 > $R/gpurun_out/r05/summarise.log 2>&1
echo "summarise rc=$?" >> $R/gpurun_out/r05/log.txt
rm -rf $R/gpurun_out/r05/stats_* $R/gpurun_out/r05/pmc_* $R/gpurun_out/r05/shape_map
du -sh $R/gpurun_out
tail -3 $R/gpurun_out/r05/pytest_gpu.txt | cut -c1-200; cat $R/gpurun_out/r05/log.txt; tail -3 $R/gpurun_out/r05/smoke.txt; ls $R/gpurun_out/r05 | head -80
