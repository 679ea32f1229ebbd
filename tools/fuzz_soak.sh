#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/fuzz_soak
mkdir -p $OUT
cd $R
(echo "python tools/fuzz_parity.py --seconds 150 --seed 71"; timeout 400 python tools/fuzz_parity.py --seconds 150 --seed 71 2>&1 | grep -v amdgpu | tail -3) > $OUT/a.txt
(echo "python tools/fuzz_parity.py --seconds 60 --seed 72 --big 0.2"; timeout 300 python tools/fuzz_parity.py --seconds 60 --seed 72 --big 0.2 2>&1 | grep -v amdgpu | tail -3) > $OUT/b.txt
(echo "RNNT_DEBUG_LATTICE_KERNEL=wl python tools/fuzz_parity.py --seconds 60 --seed 76"; RNNT_DEBUG_LATTICE_KERNEL=wl timeout 300 python tools/fuzz_parity.py --seconds 60 --seed 76 2>&1 | grep -v amdgpu | tail -3) > $OUT/b2.txt
# three at once
(timeout 400 python tools/fuzz_parity.py --seconds 90 --seed 73 2>&1 | grep -v amdgpu | tail -2 > $OUT/c1.txt) &
(RNNT_DEBUG_LATTICE_KERNEL=ws timeout 400 python tools/fuzz_parity.py --seconds 90 --seed 74 --big 0.3 2>&1 | grep -v amdgpu | tail -2 > $OUT/c2.txt) &
(timeout 400 python tools/fuzz_parity.py --seconds 90 --seed 75 --big 0.5 2>&1 | grep -v amdgpu | tail -2 > $OUT/c3.txt) &
wait
cat $OUT/a.txt $OUT/b.txt $OUT/b2.txt $OUT/c1.txt $OUT/c2.txt $OUT/c3.txt
