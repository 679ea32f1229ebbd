"""A bounded, seeded slice of tools/fuzz_parity.py inside the GPU suite: random shapes (up to U = 530 / 1100 and
V = 17000), ragged lengths, random blank, FastEmit weights, label == blank collisions, through four entry points
(`_C.rnnt_loss` dense, `rnnt_loss(gather=True)` + backward, `rnnt_loss_from_logits` + backward,
`rnnt_loss(compact=True)` + backward), each against the fp32 oracle (re-judged against fp64 where fp32 itself is
too noisy).  One run per kernel pin, so that every round-end test run walks a few hundred random cases through
the default routing and two pinned kernels; the open-ended runs are recorded in profiles/r0x_fuzz_parity.txt."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("kernel,seed", [("", 101), ("ws", 102), ("wl", 103)])
def test_random_cases_against_the_oracle(kernel, seed):
    env = dict(os.environ)
    env.pop("RNNT_DEBUG_LATTICE_KERNEL", None)
    if kernel:
        env["RNNT_DEBUG_LATTICE_KERNEL"] = kernel
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "--seconds", "6", "--seed",
                          str(seed)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = out.stdout.decode()
    assert out.returncode == 0 and "fuzz ok:" in text, text[-3000:]


@pytest.mark.parametrize("name,env_extra", [("default", {}), ("wl", {"RNNT_DEBUG_LATTICE_KERNEL": "wl"}),
                                            ("k8", {"RNNT_WD_K16_FROM_T": "1000000"})])
def test_six_processes_on_one_gpu_get_the_reference_bits_every_launch(name, env_extra):
    """tools/wd_soak.py: six processes launch the loss entry back to back on the soak's shape set at the same time (45 s each
    leg) and compare every launch -- costs, gradients, alpha and beta planes -- bit for bit with a reference computed once
    by the OTHER kernel (k_lattice_ws: compiler-scheduled, no in-place reloads, no hand-over through L2).  This is the load
    that found what no single-process test could: the hand-written lattice blocks with reloads left in flight across the
    barrier (round 5) gave wrong costs in 1-3 % of the launches under it and never otherwise, the storer's dry run without
    its wait one wrong plane in 40 000 ... 170 000 (csrc/lattice_step.h, wait_lds).  Legs: the default routes (k_lattice_wd
    with its L2 hand-over -- lost hand-overs may be redone, the bits must not change --, k_lattice_wl for the narrower
    lattices, the plain launch for single column blocks; blocks of 16 diagonals from T >= 1024, the default since round 6),
    k_lattice_wl pinned wherever it fits, and k_lattice_wd with blocks of 8 diagonals everywhere.  The static half of the gate is in the build (warp_rnnt_amd/_isa_check.py)."""
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wd_soak.py"), "--seconds", "45", "--procs", "6"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    text = out.stdout.decode()
    assert out.returncode == 0 and text.count("results differing from the k_lattice_ws reference: 0;") == 6, text[-3000:]
