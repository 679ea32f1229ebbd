cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for rep in 1 2; do
for sh in 16,1500,300 16,1500,64 16,1500,512; do
  for v in r02 hard main; do
    RNNT_LATTICE=pd python tools/lattice_probe.py --shape $sh $v: 2>&1 | grep median | sed "s/^/N,T,U=$sh lattice=pd  /"
  done
done
done > gpurun_out/r03/lattice_probe2.txt
cat gpurun_out/r03/lattice_probe2.txt
