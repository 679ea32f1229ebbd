cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wrapper.py -q -x 2>&1 | tail -5
for rep in 1 2; do
for v in new old; do
  unset RNNT_LSM_NO_REGS
  if [ $v = old ]; then export RNNT_LSM_NO_REGS=1; fi
  python bench.py --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().splitlines()[0]); print('$v', d['ms_per_step'], 'fused_fwd', d['fused_from_logits_ms'], 'train_fused', d['train_step_fused_logits_ms'], 'train_native_chain', d['train_step_native_log_softmax_chain_ms'])"
done
done
