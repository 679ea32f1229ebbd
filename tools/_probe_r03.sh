cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in "" _gather_reverse; do
  WARP_RNNT_AMD_LIB=$GRAFT_REPO_ROOT/warp_rnnt_amd/libwarp_rnnt_amd$v.so python bench.py --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().splitlines()[0]); print('lib$v', d['ms_per_step'], 'lsm', d['roofline']['kernel_ms'], 'loss', d['roofline_loss_path']['kernels_ms'], 'gather-alone', d['roofline_gather']['kernel_ms'])"
done
done
