cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do
for v in old nt0 nt1 nt2 nt3; do
  unset RNNT_LSM_NO_REGS RNNT_LSM_REGS_NT
  case $v in old) export RNNT_LSM_NO_REGS=1;; nt0) export RNNT_LSM_REGS_NT=0;; nt1) export RNNT_LSM_REGS_NT=1;; nt2) export RNNT_LSM_REGS_NT=2;; nt3) export RNNT_LSM_REGS_NT=3;; esac
  python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().splitlines()[0]); print('$v', d['ms_per_step'], 'lsm', d['roofline']['kernel_ms'], 'loss', d['roofline_loss_path']['kernels_ms'])"
done
done
