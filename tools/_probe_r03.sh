cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in "" _lgfused; do
  WARP_RNNT_AMD_LIB=$GRAFT_REPO_ROOT/warp_rnnt_amd/libwarp_rnnt_amd$v.so python bench.py --no-cpu-baseline --config c3 --steps 50 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().splitlines()[0]); print('c3 lib$v', d['ms_per_step'], 'fused_fwd', d['fused_from_logits_ms'], 'train_fused', d['train_step_fused_logits_ms'], 'train_native', d['train_step_native_log_softmax_chain_ms'])"
done
done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wrapper.py -q 2>&1 | tail -3
