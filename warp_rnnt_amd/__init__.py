"""MI355X-native RNN-Transducer loss: HIP kernels + C ABI + thin torch plumbing.

Layout
  csrc/            hand-written HIP kernels for gfx950 and the extern "C" entry points
  _build.py        hipcc build of libwarp_rnnt_amd.so (in-tree)
  _lib.py          ctypes loader for the C ABI declared in include/warp_rnnt_amd.h
  ops.py           torch-tensor front ends of the native entry points
  functional.py    log_softmax whose result rnnt_loss(..., gather=True) recognises and fuses with (lazy)
  debug.py         kernel pin for A/B runs, last_lattice_kernel()
  distributed.py   batch-sharded loss over RCCL (one rank per GPU)

The drop-in package that mirrors the reference's Python interface is the
top-level ``warp_rnnt`` (``warp_rnnt.rnnt_loss``, ``warp_rnnt._C.rnnt_loss``).
There is no CPU fallback anywhere in these packages.
"""
from ._lib import load, lib_path, RNNTStatusError  # noqa: F401
from ._mismatch import last_mismatch  # noqa: F401

__version__ = "0.2.0"
