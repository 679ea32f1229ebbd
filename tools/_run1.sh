set -u
mkdir -p gpurun_out/r06a
python -m pytest tests -m gpu -x -q > gpurun_out/r06a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06a/pytest.log
for leg in "default:" "wl:RNNT_LOGDOMAIN_KERNEL=wl" "k16:RNNT_WD_K16_FROM_T=1024"; do
  name=${leg%%:*}; envs=${leg#*:}
  echo "== 6 procs, $name ($envs)" >> gpurun_out/r06a/soak.txt
  env $envs python tools/wd_soak.py --seconds 60 --procs 6 >> gpurun_out/r06a/soak.txt 2>&1
done
echo "== 3 procs, default" >> gpurun_out/r06a/soak.txt
python tools/wd_soak.py --seconds 60 --procs 3 >> gpurun_out/r06a/soak.txt 2>&1
python bench.py --no-cpu-baseline > gpurun_out/r06a/bench_c4.json 2> gpurun_out/r06a/bench_c4.err
RNNT_WD_K16_FROM_T=1024 python bench.py --no-cpu-baseline > gpurun_out/r06a/bench_c4_k16.json 2>> gpurun_out/r06a/bench_c4.err
tail -3 gpurun_out/r06a/pytest.log; cat gpurun_out/r06a/soak.txt | grep -v "^  " | tail -30
