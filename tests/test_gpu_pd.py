"""The probability-domain lattice kernel is opt-in (warp_rnnt_amd.set_lattice("pd"); csrc/lattice.hip: launch_lattice):
here it runs everything it supports (a subprocess because the second test loads another build of the library) and is
compared with the fp32 oracle.  The default route (the reference's log-domain arithmetic) is what the other GPU tests
exercise; c4 and c5 of tests/test_gpu_baseline_sizes.py run on both."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_forced_probability_domain_kernel_against_oracle():
    env = dict(os.environ)
    out = subprocess.run([sys.executable, os.path.join(HERE, "pd_vs_oracle.py")], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=900)
    text = out.stdout.decode()
    assert out.returncode == 0 and "PD_VS_ORACLE_OK" in text, text[-4000:]


def test_lost_hand_over_falls_back_to_log_domain_kernel():
    """A hand-over wait is bounded; when it gives up, the sweep carries on with whatever the ring holds, is flagged
    and redone by the log-domain kernel behind. The `short_spin` build gives up at the first poll, so every column
    block that catches up with its neighbour takes that path; results must not change."""
    from warp_rnnt_amd import _build
    lib = _build.build(variant="short_spin")      # (a no-op when the variant built by __graft_entry__.build() is current)
    env = dict(os.environ, WARP_RNNT_AMD_LIB=lib, PD_VS_ORACLE_EXPECT_REDO="1")
    out = subprocess.run([sys.executable, os.path.join(HERE, "pd_vs_oracle.py")], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=900)
    text = out.stdout.decode()
    assert out.returncode == 0 and "PD_VS_ORACLE_OK" in text, text[-4000:]
