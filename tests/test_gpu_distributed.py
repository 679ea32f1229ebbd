"""The batch-sharded loss with REAL kernels in more than one process.

No 8-GPU node is available to the build, and RCCL refuses two ranks on one device, so the N>1 path is exercised
here as two processes that share cuda:0 and talk over gloo: each rank takes its `shard_batch` slice of one seeded
global minibatch, runs `sharded_rnnt_loss` (log-softmax, gather, lattice, gradients: the HIP kernels) and
`backward()`; the global loss of every reduction and every rank's gradient slice are checked against the fp32
oracle run on the UNSHARDED batch.  What this pins that the one-rank tests cannot: utterances really are
independent across processes (same bits as the single-process run of the whole batch), the scalar exchange carries
the right normalisation for 'mean' with uneven shards (5 utterances on 2 ranks: 3 + 2), and nothing in the library
(the per-device launch counter, the hand-over rings, the sticky diagnostics words) is confused by a second process on
the same device.

The reference has no counterpart (no collective call site exists)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import oracle
from helpers import make_case, np_log_softmax32
from warp_rnnt_amd import ops
from warp_rnnt_amd.distributed import shard_batch, shard_bounds, sharded_rnnt_loss
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda:0")                      # both ranks on the one GPU of the test box
N, T, U, V = {shape}
logits, labels, xn, yn = make_case(31, N, T, U, V, ragged=True)
lp_cpu = np_log_softmax32(logits)
full = oracle.rnnt_loss_f32(lp_cpu, labels, xn, yn, fastemit_lambda=0.01)
lo, hi = shard_bounds(N, rank, world)
g_logits, g_labels, g_xn, g_yn = (torch.tensor(a, device=dev) for a in (logits, labels, xn, yn))
my = shard_batch((g_logits, g_labels, g_xn, g_yn), rank, world)
tol = dict(rtol=1e-5, atol=1e-4 * max(1.0, float(np.abs(full["costs"]).max()) / 100.0))
for red in ("mean", "sum", "none"):
    for gather in (False, True):
        lp = ops.log_softmax(my[0]).requires_grad_(True)
        loss, glob = sharded_rnnt_loss(lp, my[1], my[2], my[3], reduction=red, gather=gather, fastemit_lambda=0.01,
                                       n_global=N)
        if red == "none":
            np.testing.assert_allclose(glob.cpu().numpy(), full["costs"], rtol=1e-5)
            np.testing.assert_allclose(loss.detach().cpu().numpy(), full["costs"][lo:hi], rtol=1e-5)
            loss.sum().backward(); scale = 1.0
        else:
            want = full["costs"].sum() if red == "sum" else full["costs"].mean()
            np.testing.assert_allclose(glob.item(), want, rtol=1e-5)
            # this rank's share of the global objective; summed over ranks it is the global loss
            t = loss.detach().clone(); dist.all_reduce(t)
            np.testing.assert_allclose(t.item(), want, rtol=1e-5)
            loss.backward(); scale = 1.0 if red == "sum" else 1.0 / N
        # d(global loss)/d(this rank's log-probs) = the oracle's gradient of the unsharded batch, this rank's rows
        np.testing.assert_allclose(lp.grad.cpu().numpy(), full["grads"][lo:hi] * scale, **tol)
# a second process on the device must not disturb the first: both now run the ring kernel (k_lattice_wd: three column
# blocks handing over through L2, its per-device launch counter, its flags and rings) at the same time on their own workspaces
logits, labels, xn, yn = make_case(32 + rank, 2, 700, 150, 7, ragged=True)
ref = oracle.rnnt_loss_f32(np_log_softmax32(logits), labels, xn, yn)
dist.barrier()
for _ in range(3):
    lp = ops.log_softmax(torch.tensor(logits, device=dev)).requires_grad_(True)
    loss, glob = sharded_rnnt_loss(lp, *(torch.tensor(a, device=dev) for a in (labels, xn, yn)), reduction="sum",
                                   gather=True)
    loss.backward()
    np.testing.assert_allclose(loss.item(), ref["costs"].sum(), rtol=1e-5)
    # (|log-likelihood| ~ 1e3 here: the fp32 ORACLE is a few 1e-3 from exact arithmetic, the kernel is not)
    np.testing.assert_allclose(lp.grad.cpu().numpy(), ref["grads"], atol=5e-3)
torch.cuda.synchronize()
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("shape", [(5, 40, 12, 9), (4, 150, 70, 28)])
def test_two_ranks_share_one_gpu_real_kernels(tmp_path, shape):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, shape=shape))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    report = "\n".join(f"---- rank {r} (exit {p.returncode}) ----\n{o[-3000:]}" for r, (p, o) in enumerate(zip(procs, outs)))
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, report
