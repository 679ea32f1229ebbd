"""A bounded, seeded slice of tools/fuzz_parity.py inside the GPU suite: random shapes (up to U = 530 / 1100 and
V = 17000), ragged lengths, random blank, FastEmit weights, label == blank collisions, through four entry points
(`_C.rnnt_loss` dense, `rnnt_loss(gather=True)` + backward, `rnnt_loss_from_logits` + backward,
`rnnt_loss(compact=True)` + backward), each against the fp32 oracle (re-judged against fp64 where fp32 itself is
too noisy).  One run per lattice route, so that every round-end test run walks a few hundred random cases through
both arithmetics; the open-ended runs are recorded in profiles/r0x_fuzz_parity.txt."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("route,seed", [("auto", 101), ("pd", 102), ("logdomain", 103)])
def test_random_cases_against_the_oracle(route, seed):
    env = dict(os.environ, RNNT_LATTICE=route if route != "auto" else "")
    if route == "auto":
        env.pop("RNNT_LATTICE")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "--seconds", "6", "--seed",
                          str(seed)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = out.stdout.decode()
    assert out.returncode == 0 and "fuzz ok:" in text, text[-3000:]
