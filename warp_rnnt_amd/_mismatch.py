"""Host side of the sticky forward/backward-mismatch diagnostics (include/warp_rnnt_amd.h: rnnt_amd_mismatch_flag).

The reference prints from the device whenever its alpha/beta consistency guard fires ("WARNING: sample %d [%d, %d] has a
forward/backward mismatch %f / %f", core_gather.cu:345-349) -- the only diagnostic it has for a lattice that went wrong.
Here the gradient kernel writes the same facts into eight words of pinned host memory per device, only when a guard
fires; this module looks at those words around every loss call and in every backward -- a plain memory read, no device
synchronisation, nothing when nothing fired -- and turns a firing into one ``RuntimeWarning``.  Because nothing waits for
the GPU, the warning appears at the first look AFTER the kernel that fired has run: normally the next step's call, or the
backward of the same step.  ``warp_rnnt_amd.last_mismatch()`` returns the details of the last one seen.

``WARP_RNNT_AMD_CHECK_MISMATCH``: unset = this (default on); ``off`` = never look; ``warn`` / ``raise`` = additionally
read the per-call flags back after every call (one host synchronisation per call: exact, and immediate)."""
import os
import struct
import warnings

from . import _lib

_words = {}          # device index -> ctypes pointer to the device's eight words (None: could not be set up)
_last = None
_count = 0


def _off():
    return os.environ.get("WARP_RNNT_AMD_CHECK_MISMATCH", "").lower() in ("off", "0", "no")


def _f32(bits):
    return struct.unpack("<f", struct.pack("<I", bits & 0xffffffff))[0]


def poll(device, stacklevel=3):
    """Look at `device`'s words (a torch.device or an index); warn once per firing.  Returns the details or None."""
    global _last, _count
    if _off():
        return None
    idx = device if isinstance(device, int) else (device.index if device.index is not None else _current())
    w = _words.get(idx)
    if w is None:
        if idx in _words:
            return None                  # could not be set up
        w = _setup(idx)
        if w is False:
            return None                  # inside a stream capture: try again at the next call outside one
        _words[idx] = w
        if w is None:
            return None
    if not w[0]:
        return None
    kind, n, xn, yn, lla, b00 = int(w[1]), int(w[2]), int(w[3]), int(w[4]), _f32(w[5]), _f32(w[6])
    w[0] = 0
    _count += 1
    _last = {"device": idx, "kind": "mismatch" if kind == 1 else "invalid lengths", "utterance": n, "frames": xn,
             "labels": yn, "loglik_alpha": lla, "loglik_beta": b00, "seen": _count}
    if kind == 1:
        msg = (f"rnnt_loss: sample {n} [{xn}, {yn}] has a forward/backward mismatch {lla:f} / {b00:f} (cuda:{idx}): its "
               "gradients were zeroed and its cost is the mean of the two, as in the reference (core_gather.cu:341-354). "
               "Seen without synchronising: the call that fired is this one or an earlier one; "
               "warp_rnnt_amd.last_mismatch() has the details")
    else:
        msg = (f"rnnt_loss: sample {n} has lengths out of range (frames {xn}, labels {yn}; cuda:{idx}): cost NaN, gradients "
               "zero (the reference reads out of bounds here, binding.cpp:47-51); warp_rnnt_amd.last_mismatch()")
    warnings.warn(msg, RuntimeWarning, stacklevel=stacklevel)
    return _last


def _current():
    import torch
    return torch.cuda.current_device()


def _setup(idx):
    """First look at a device: ask the library for its words (allocates the table once per process).  Not inside a stream
    capture (the allocation is not capturable): the device stays unwatched until the next call outside one."""
    import torch
    try:
        if torch.cuda.is_current_stream_capturing():
            return False
    except RuntimeError:
        return None
    p = _lib.load().rnnt_amd_mismatch_flag(int(idx))
    return p if p else None


def last_mismatch():
    """Details of the last forward/backward mismatch (or invalid-length utterance) this process has SEEN -- a dict with
    device, kind, utterance, frames, labels, loglik_alpha, loglik_beta, seen (how many were seen so far) -- or None.
    Looks at every watched device first; call ``torch.cuda.synchronize()`` before it for an exact answer."""
    for idx in list(_words):
        poll(idx, stacklevel=3)
    return _last
