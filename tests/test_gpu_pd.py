"""The probability-domain lattice kernel is chosen automatically only for long lattices of small batches
(csrc/lattice.hip: launch_lattice); here it is forced on for everything it supports (RNNT_LATTICE=pd is read
once per process, hence the subprocess) and compared with the fp32 oracle.  The default routing is what the
other GPU tests exercise: short lattices run the log-domain kernels, c4 and c5 of
tests/test_gpu_baseline_sizes.py the probability-domain one."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_forced_probability_domain_kernel_against_oracle():
    env = dict(os.environ, RNNT_LATTICE="pd")
    out = subprocess.run([sys.executable, os.path.join(HERE, "pd_vs_oracle.py")], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=900)
    text = out.stdout.decode()
    assert out.returncode == 0 and "PD_VS_ORACLE_OK" in text, text[-4000:]
