#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r06
cd $R
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 --timeout 900 -p no:cacheprovider > $R/gpurun_out/r06/pytest_gpu.txt 2>&1
echo "pytest rc=$?" > $R/gpurun_out/r06/log.txt
python -c "import __graft_entry__ as g; g.smoke()" > $R/gpurun_out/r06/smoke.txt 2>&1
echo "smoke rc=$?" >> $R/gpurun_out/r06/log.txt
bash tools/collect_profiles.sh r06 > $R/gpurun_out/r06/collect.log 2>&1
echo "collect rc=$?" >> $R/gpurun_out/r06/log.txt
# the raw traces exceed what gpurun copies back: summarise here, keep the summaries and the small files
PROFILES_DST=$R/gpurun_out/r06/summary python tools/summarise_profiles.py r06 > $R/gpurun_out/r06/summarise.log 2>&1
echo "summarise rc=$?" >> $R/gpurun_out/r06/log.txt
rm -rf $R/gpurun_out/r06/stats_* $R/gpurun_out/r06/pmc_* $R/gpurun_out/r06/shape_map
bash tools/gather_store_pmc.sh r06 > $R/gpurun_out/r06/gather_store_pmc.log 2>&1
du -sh $R/gpurun_out
tail -3 $R/gpurun_out/r06/pytest_gpu.txt | cut -c1-200; cat $R/gpurun_out/r06/log.txt; tail -3 $R/gpurun_out/r06/smoke.txt; ls $R/gpurun_out/r06 | head -80
