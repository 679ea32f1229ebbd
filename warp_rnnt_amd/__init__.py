"""MI355X-native RNN-Transducer loss: HIP kernels + C ABI + thin torch plumbing.

Layout
  csrc/            hand-written HIP kernels for gfx950 and the extern "C" entry points
  _build.py        hipcc build of libwarp_rnnt_amd.so (in-tree)
  _lib.py          ctypes loader for the C ABI declared in include/warp_rnnt_amd.h
  ops.py           torch-tensor front ends of the native entry points
  distributed.py   batch-sharded loss over RCCL (one rank per GPU)

The drop-in package that mirrors the reference's Python interface is the
top-level ``warp_rnnt`` (``warp_rnnt.rnnt_loss``, ``warp_rnnt._C.rnnt_loss``).
There is no CPU fallback anywhere in these packages.
"""
import contextlib

from ._lib import load, lib_path, RNNTStatusError  # noqa: F401

LATTICE_ROUTES = ("auto", "logdomain", "pd")


def set_lattice(route):
    """Select the arithmetic of the alpha/beta sweeps for every later call in this process (include/warp_rnnt_amd.h,
    ``rnnt_amd_set_lattice``): ``"auto"`` (default) and ``"logdomain"`` are the reference's fp32 log-sum-exp per cell
    -- the same bits whatever the batch an utterance is computed in; ``"pd"`` is the probability-domain kernel wherever
    it is supported (padded or compact layout, U <= 512): closer to exact arithmetic on long lattices (7e-4 instead of
    1e-2 on the gradients at T=1500, U=300), not the reference's numbers.  Process-wide: the DEFAULT of every call that
    does not name its own route -- ``warp_rnnt_amd.ops.loss(..., lattice="pd")`` / ``ops.loss_compact(..., lattice=)``
    choose per call and touch no state, which is what callers on several threads want.  Returns the previous route."""
    if route not in LATTICE_ROUTES:
        raise ValueError(f"unknown lattice route {route!r}: expected one of {LATTICE_ROUTES}")
    return LATTICE_ROUTES[load().rnnt_amd_set_lattice(LATTICE_ROUTES.index(route))]


def get_lattice():
    return LATTICE_ROUTES[load().rnnt_amd_get_lattice()]


@contextlib.contextmanager
def lattice_route(route):
    """``with warp_rnnt_amd.lattice_route("pd"): ...`` -- the route inside the block, the old one after it (process-wide:
    every thread's calls see it while the block runs)."""
    old = set_lattice(route)
    try:
        yield
    finally:
        set_lattice(old)

LOGDOMAIN_KERNELS = ("auto", "ws", "wd", "wl")


def set_logdomain_kernel(kernel):
    """Which kernel serves the log-domain arithmetic (``rnnt_amd_set_logdomain_kernel``): ``"auto"`` by shape, ``"ws"``
    one workgroup per sweep, ``"wd"`` one workgroup per 64-column block, ``"wl"`` the single-workgroup form of ``wd``
    wherever it fits.  Same bits whichever runs: a tuning / test knob.
    Process-wide, not thread-safe (like :func:`set_lattice`).  Returns the previous setting."""
    if kernel not in LOGDOMAIN_KERNELS:
        raise ValueError(f"unknown log-domain kernel {kernel!r}: expected one of {LOGDOMAIN_KERNELS}")
    return LOGDOMAIN_KERNELS[load().rnnt_amd_set_logdomain_kernel(LOGDOMAIN_KERNELS.index(kernel))]


LATTICE_KERNELS = ("none", "lattice_ws", "lattice_wd", "lattice_pd", "lattice (single role)", "lattice_wl")


def last_lattice_kernel():
    """Name of the lattice kernel this thread's last loss call launched (``rnnt_amd_debug_last_lattice_kernel``)."""
    return LATTICE_KERNELS[load().rnnt_amd_debug_last_lattice_kernel()]


__version__ = "0.1.0"
