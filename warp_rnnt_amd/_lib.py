"""ctypes binding of the C ABI in include/warp_rnnt_amd.h.

Import torch before calling :func:`load` in a process that also uses torch:
the library depends on ``libamdhip64.so.7`` and must share the HIP runtime
torch already loaded (same SONAME, so the dynamic loader reuses it), otherwise
stream handles would belong to a different runtime.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libwarp_rnnt_amd.so"
_lib = None

STATUS_NAMES = {
    0: "RNNT_STATUS_SUCCESS", 1: "RNNT_STATUS_WARP_FAILED", 2: "RNNT_STATUS_GRADS_BLANK_FAILED",
    3: "RNNT_STATUS_GRADS_LABEL_FAILED", 4: "RNNT_STATUS_COSTS_FAILED",
    5: "RNNT_STATUS_INVALID_ARGUMENT", 6: "RNNT_STATUS_PROLOGUE_FAILED", 7: "RNNT_STATUS_EXPAND_FAILED",
}

# input_kind / grads_kind enums of rnnt_amd_loss
IN_LOG_PROBS_DENSE, IN_LOG_PROBS_GATHERED, IN_LOGITS_DENSE = 0, 1, 2
GRADS_GATHERED, GRADS_GATHERED_DIAGONAL, GRADS_DENSE, GRADS_NONE = 0, 1, 2, 3

# every symbol include/warp_rnnt_amd.h declares: (restype, argtypes)
_vp, _i, _f, _sz, _i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_int64
SYMBOLS = {
    "run_warp_rnnt": (_i, [_vp] * 10 + [_i] * 5 + [_f]),
    "run_warp_rnnt_gather": (_i, [_vp] * 9 + [_i] * 3 + [_f]),
    "run_gather_for_compact": (None, [_vp] * 8 + [ctypes.c_uint] * 5),
    "run_warp_rnnt_compact": (None, [_vp] * 10 + [ctypes.c_uint] * 3 + [_f, ctypes.c_bool]),
    "run_scatter_grad_for_compact": (None, [_vp] * 5 + [ctypes.c_uint] * 4),
    "rnnt_amd_compact_last_status": (_i, []),
    "rnnt_amd_workspace_size": (_sz, [_i, _i, _i]),
    "rnnt_amd_workspace_mismatch_offset": (_sz, [_i, _i, _i]),
    "rnnt_amd_debug_redo_offset": (_sz, [_i, _i, _i]),
    "rnnt_amd_loss": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f]),
    "rnnt_amd_expand_grads": (_i, [_vp] * 7 + [_i] * 6),
    "rnnt_amd_logits_backward": (_i, [_vp] * 6 + [_i] * 5),
    "rnnt_amd_log_softmax": (_i, [_vp, _vp, _vp, _i64, _i]),
    "rnnt_amd_log_softmax_backward": (_i, [_vp, _vp, _vp, _vp, _i64, _i]),
    "rnnt_amd_gather": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "rnnt_amd_workspace_size_compact": (_sz, [_i, _i64, _i, _i]),
    "rnnt_amd_loss_compact": (_i, [_vp] * 11 + [_i, _i64, _i, _i, _i, _i, _f]),
    "rnnt_amd_workspace_size_compact_bounded": (_sz, [_i, _i64, _i, _i]),
    "rnnt_amd_loss_compact_bounded": (_i, [_vp] * 4 + [_i64] + [_vp] * 5 + [_i, _i64, _i, _i, _i, _i, _f]),
    "rnnt_amd_compact_scatter_grads": (_i, [_vp] * 6 + [_i64, _i, _i, _i]),
    "rnnt_amd_compact_offsets": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "rnnt_amd_debug_lattice_only": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i]),
    "rnnt_amd_debug_gather_only": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "rnnt_amd_debug_set_lattice_kernel": (_i, [_i]),
    "rnnt_amd_debug_get_lattice_kernel": (_i, []),
    "rnnt_amd_mismatch_flag": (ctypes.POINTER(ctypes.c_uint), [_i]),
    "rnnt_amd_debug_last_lattice_kernel": (_i, []),
    "rnnt_amd_version": (_i, []),
}


ABI_VERSION = 106   # rnnt_amd_version() of the library these argument lists belong to


class RNNTStatusError(RuntimeError):
    pass


def lib_path():
    """The in-tree library; WARP_RNNT_AMD_LIB selects another build of the same C ABI (the A/B variants
    of _build.VARIANTS used for the parity table) -- still a HIP build, still no fallback."""
    return os.environ.get("WARP_RNNT_AMD_LIB") or os.path.join(HERE, _LIB_NAME)


def load():
    """Load libwarp_rnnt_amd.so; fail loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `python warp_rnnt_amd/_build.py`). "
            "There is no CPU fallback.")
    L = ctypes.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)   # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    # the argument lists above are this version's: an older build of the library (a stale prebuilt .so, a
    # WARP_RNNT_AMD_LIB variant built from older sources) would accept the calls and misread them
    if L.rnnt_amd_version() != ABI_VERSION:
        raise RuntimeError(f"{path} reports C-ABI version {L.rnnt_amd_version()}, this package binds version "
                           f"{ABI_VERSION}: rebuild it (`python warp_rnnt_amd/_build.py`)")
    _lib = L
    return L
