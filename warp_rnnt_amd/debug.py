"""Diagnostics and A/B knobs -- nothing here is needed in use and nothing in the product calls it.

One arithmetic serves every call (the reference's fp32 log-sum-exp per cell); several kernels implement it with the same
instructions on the chain and the same bits, and the library picks one by shape.  `set_lattice_kernel` pins one for tests
and timing runs (process-wide: `rnnt_amd_debug_set_lattice_kernel`, one atomic int; initial value from the environment
variable RNNT_DEBUG_LATTICE_KERNEL = ws | wd | wl)."""
import contextlib

from ._lib import load

LATTICE_KERNEL_PINS = ("auto", "ws", "wd", "wl")
LATTICE_KERNELS = ("none", "lattice_ws", "lattice_wd", "(retired)", "lattice (single role)", "lattice_wl")


def set_lattice_kernel(kernel):
    """``"auto"`` by shape, ``"ws"`` compute + I/O wave pairs in one workgroup per sweep, ``"wd"`` one three-wave
    workgroup per 64-column block (boundary columns through L2), ``"wl"`` the same teams in one workgroup per sweep
    wherever it fits.  Same bits whichever runs.  Returns the previous pin."""
    if kernel not in LATTICE_KERNEL_PINS:
        raise ValueError(f"unknown lattice kernel {kernel!r}: expected one of {LATTICE_KERNEL_PINS}")
    return LATTICE_KERNEL_PINS[load().rnnt_amd_debug_set_lattice_kernel(LATTICE_KERNEL_PINS.index(kernel))]


def get_lattice_kernel():
    return LATTICE_KERNEL_PINS[load().rnnt_amd_debug_get_lattice_kernel()]


@contextlib.contextmanager
def lattice_kernel(kernel):
    """``with debug.lattice_kernel("ws"): ...`` -- the pin inside the block, the old one after it (process-wide)."""
    old = set_lattice_kernel(kernel)
    try:
        yield
    finally:
        set_lattice_kernel(old)


def last_lattice_kernel():
    """Name of the lattice kernel this thread's last loss call launched (``rnnt_amd_debug_last_lattice_kernel``)."""
    return LATTICE_KERNELS[load().rnnt_amd_debug_last_lattice_kernel()]
