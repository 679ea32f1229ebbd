"""The library the timing probes under tools/ run: the `ab` build variant (warp_rnnt_amd/_build.py: -DRNNT_AB_KNOBS), the
one build that reads the kernel-selection knobs of DESIGN.md section 10 from the environment.  The shipped library
ignores those variables (csrc/common.h: ab_getenv), so a probe that A/Bs a knob has to load this one; results are the
same bits either way.  Build it in the build container (`python warp_rnnt_amd/_build.py ab`): it travels to the GPU box
with the tree like the shipped library."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def use_ab_build():
    if os.environ.get("WARP_RNNT_AMD_LIB"):
        return os.environ["WARP_RNNT_AMD_LIB"]
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from warp_rnnt_amd import _build
    os.environ["WARP_RNNT_AMD_LIB"] = _build.build(variant="ab")
    return os.environ["WARP_RNNT_AMD_LIB"]
