cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
(echo "== RNNT_LATTICE=pd --seconds 120 --seed 61"; RNNT_LATTICE=pd timeout 300 python tools/fuzz_parity.py --seconds 120 --seed 61 2>&1 | tail -3
echo "== RNNT_LATTICE=pd --seconds 60 --seed 62 --big 0.3"; RNNT_LATTICE=pd timeout 300 python tools/fuzz_parity.py --seconds 60 --seed 62 --big 0.3 2>&1 | tail -3
echo "== default routing --seconds 90 --seed 63"; timeout 300 python tools/fuzz_parity.py --seconds 90 --seed 63 2>&1 | tail -3) > gpurun_out/r03/fuzz1.txt
cat gpurun_out/r03/fuzz1.txt
python bench.py --no-cpu-baseline > gpurun_out/r03/bench3.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/bench3.json').read().splitlines()[0])
print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline_loss_path']['kernels_ms'], d.get('roofline_gather'))
PY
