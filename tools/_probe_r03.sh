cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for r in auto logdomain; do
  RNNT_LATTICE=$r timeout 300 python tools/benchmark_table.py --loss warp-rnnt-compact --random_length 2>&1 | tail -22 > gpurun_out/r03/table_compact_$r.txt
done
paste gpurun_out/r03/table_compact_auto.txt gpurun_out/r03/table_compact_logdomain.txt | cut -c1-250
