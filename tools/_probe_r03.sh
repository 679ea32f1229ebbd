cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
(timeout 700 python -m pytest tests/test_gpu_compact.py tests/test_gpu_pd.py -q 2>&1 | tail -25) > gpurun_out/r03/pytest_compact.log
cat gpurun_out/r03/pytest_compact.log | cut -c1-400
for l in warp-rnnt-compact; do
  for r in auto logdomain; do
    RNNT_LATTICE=$r timeout 300 python tools/benchmark_table.py --loss $l --random_length True 2>&1 | tail -25 > gpurun_out/r03/table_compact_$r.txt
  done
done
paste gpurun_out/r03/table_compact_auto.txt gpurun_out/r03/table_compact_logdomain.txt | cut -c1-250
