"""Builds libwarp_rnnt_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU.  The shared object lands next to this file
so that it travels with the source tree (it is git-ignored, not packaged).
"""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

from . import _isa_check

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libwarp_rnnt_amd.so")
SOURCES = ["api.hip", "lattice.hip", "lattice_ws.hip", "lattice_wd.hip", "grads.hip", "prologue.hip", "expand.hip"]
HEADERS = ["common.h", "kernels.h", "lattice_step.h", "lattice_wd_body.h", "lattice_single.h", "grads_cell.h",
           os.path.join("..", "..", "include", "warp_rnnt_amd.h")]
ARCH = "gfx950"
# Sources whose kernels refill live registers with inline-assembly LDS loads the compiler does not count (lattice_step.h):
# the ISA of the object that ships is checked by _isa_check on EVERY build, and a violation fails the build (round 5
# shipped two silent wrong-answer bugs of this family).  Value: (kernels, in-place reloads) the file is known to hold at
# least -- a check that no longer finds them must not pass.
RELOAD_CHECKED = {"lattice_wd.hip": (8, 600)}
# Sources with inline assembly of any kind: their ISA is walked for the wait-state hazards the compiler's recognizer does
# not resolve around an `asm` statement (_isa_check: third rule -- DPP behind a VALU write, a transcendental's result in
# the next slot, a VALU write behind a wide store).  Value: kernels the file is known to hold at least.
HAZARD_CHECKED = {"lattice_wd.hip": 8, "lattice.hip": 4, "lattice_ws.hip": 2, "prologue.hip": 60}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def _fingerprint(files, flags):
    import hashlib
    h = hashlib.sha256(" ".join(flags).encode())
    for p in files:
        with open(p, "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    return h.hexdigest()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# A/B builds for the parity table (tests/test_gpu_baseline_sizes.py, profiles/r02_parity_errors.json);
# select one at run time with WARP_RNNT_AMD_LIB=<path> (see _lib.lib_path).
VARIANTS = {
    # the single-role kernel with ocml expf/log1pf in the chain -- the reference's own lse (core.cu:26-39) bit for bit, 4x slower
    "precise": ["-DRNNT_LATTICE_LEGACY", "-DRNNT_PRECISE_LIBM"],
    # hand-over waits that give up at once: every column block that catches up with its neighbour flags its sweep
    # for the kernel behind (the "producer lost" path, which never triggers otherwise)
    "short_spin": ["-DRNNT_WD_SPIN_LIMIT=0"],
    # A/B of k_lattice_wl's wave placement: six waves in column-block-major order / compute waves at raised priority
    "wl_nopad": ["-DRNNT_WL_PAD=0"],
    "wl_noprio": ["-DRNNT_WL_PRIO=0"],
    # the build that reads the kernel-selection knobs of DESIGN.md section 10 from the environment (common.h: ab_getenv); the
    # probes under tools/ load it through WARP_RNNT_AMD_LIB -- the shipped library ignores those variables
    "ab": ["-DRNNT_AB_KNOBS"],
    # A/B: the dense gather's pair stores left dirty in L2 (rounds 1-5) instead of written through (sc1)
    "gather_plain_stores": ["-DRNNT_GATHER_STORE_SC1=0"],
    # A/B: the blocks of k_lattice_wd / k_lattice_wl in which lanes finish in the predicated C++ form (rounds 4-6) instead of
    # the hand-written steady-state code
    "wd_masked_tail": ["-DRNNT_WD_FAST_TAIL=0"],
    # A/B: the register log-softmax kernel's row maxima by fmaxf() on DPP results (rounds 3-5) instead of v_max_f32_dpp
    "lsm_regs_c_max": ["-DRNNT_LSM_REGS_ASM_MAX=0"],
    # A/B: ... its results stored straight from the registers (400-byte segments at V=50) instead of in address order via LDS
    "lsm_regs_direct": ["-DRNNT_LSM_REGS_LINEAR=0"],
    "lsm_regs_r05": ["-DRNNT_LSM_REGS_LINEAR=0", "-DRNNT_LSM_REGS_ASM_MAX=0"],
    "lsm_regs_nt_stores": ["-DRNNT_LSM_REGS_STORE_WT=0"],
    # timing probes of the fused logits -> pairs kernel's stores (WRONG results: tools/fused_store_probe.py only)
    "probe_hot_pairs": ["-DRNNT_PROBE_HOT_PAIRS"],
    "probe_linear_pairs": ["-DRNNT_PROBE_LINEAR_PAIRS"],
    # a build that MUST FAIL: the hand-written blocks end with two of their in-place reloads still in flight -- the bug
    # class of round 5; tests/test_host_cpu.py checks that build() refuses it (warp_rnnt_amd/_isa_check.py)
    "planted_violation": ["-DRNNT_PLANT_RELOAD_VIOLATION"],
}


def variant_path(variant):
    return os.path.join(HERE, f"libwarp_rnnt_amd_{variant}.so")


def build(force=False, verbose=False, extra_flags=(), variant=None):
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build" if variant is None else "build_" + variant)
    lib = LIB if variant is None else variant_path(variant)
    if variant is not None:
        extra_flags = list(VARIANTS[variant]) + list(extra_flags)
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
             "-fno-slp-vectorize"]   # the lattice chains must stay scalar (v_pk_add_f32 costs two issue slots)
    flags += list(extra_flags)

    srcs = [os.path.join(CSRC, x) for x in SOURCES]
    _isa_check.require_no_wide_asm_stores(srcs + hdrs)       # (cheap, every call: a rule on the source text)
    fp = _fingerprint(srcs + hdrs + [_isa_check.__file__], flags)     # (a changed gate re-examines what it guards)
    if not force and os.path.exists(lib) and _read(lib + ".fingerprint") == fp:
        # built from exactly these sources with exactly these flags: nothing to do, even when the object
        # files are absent and whatever the file times say (only the .so and this sidecar travel to the
        # GPU box, and a snapshot copy need not preserve mtimes)
        return lib

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        checked = RELOAD_CHECKED.get(src)
        hazard_checked = HAZARD_CHECKED.get(src)
        ok_mark = o + ".reloads_ok"
        if force or _stale(o, [s] + hdrs) or ((checked or hazard_checked) and _read(ok_mark) != fp):
            # (the files with hand-placed LDS reloads are compiled through their assembly text, -save-temps=obj, so that
            #  what is checked below is what is assembled into the object -- not a second compilation's output)
            cmd = [hipcc] + flags + (["-save-temps=obj"] if (checked or hazard_checked) else []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            if os.path.exists(ok_mark):
                os.remove(ok_mark)
            subprocess.check_call(cmd)
            if checked or hazard_checked:
                isa = os.path.join(objdir, src.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-" + ARCH + ".s")
                stem = os.path.join(objdir, src.replace(".hip", ""))
                try:
                    kernels, reloads = _isa_check.require_clean(isa, *checked) if checked else (0, 0)
                    if hazard_checked:
                        hk, hi, _ = _isa_check.require_no_asm_hazards(isa, hazard_checked)
                except _isa_check.ReloadCheckError:
                    os.remove(o)          # nothing links against an object that failed its check (its ISA stays, to look at)
                    raise
                finally:                  # (-save-temps leaves ~15 MB of intermediates per file)
                    for f in os.listdir(objdir):
                        full = os.path.join(objdir, f)
                        if full.startswith(stem + "-h") and not (f.endswith(ARCH + ".s") and not os.path.exists(o)):
                            os.remove(full)
                    if os.path.exists(stem + ".hip-hip-amdgcn-amd-amdhsa.hipfb"):
                        os.remove(stem + ".hip-hip-amdgcn-amd-amdhsa.hipfb")
                if verbose and checked:
                    print(f"{src}: {kernels} lattice kernels, {reloads} in-place LDS reloads checked, 0 violations", flush=True)
                if verbose and hazard_checked:
                    print(f"{src}: {hk} kernels, {hi} instructions walked for wait-state hazards around inline assembly, 0 found",
                          flush=True)
                with open(ok_mark, "w") as f:
                    f.write(fp)
        return o

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(lib, objs) or _read(lib + ".fingerprint") != fp:
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(lib + ".fingerprint", "w") as f:
            f.write(fp)
    return lib


BINDING_SRC = os.path.join(CSRC, "binding.cpp")
BINDING = os.path.normpath(os.path.join(HERE, "..", "warp_rnnt", "_C_native.so"))


def build_binding(force=False, verbose=False):
    """Compile warp_rnnt/_C_native.so: the pybind/ATen host binding over libwarp_rnnt_amd.so (host code only, g++).
    Returns the path, or None when the toolchain pieces are missing (the ctypes module then serves)."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    lib = build()                                   # the kernels' library it links against
    hdr = os.path.normpath(os.path.join(CSRC, "..", "..", "include", "warp_rnnt_amd.h"))
    gxx = shutil.which(os.environ.get("CXX", "g++"))
    if gxx is None:
        return None
    tdir = os.path.dirname(torch.__file__)
    flags = ["-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-D__HIP_PLATFORM_AMD__=1",
             "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_C_native", "-DTORCH_API_INCLUDE_EXTENSION_H",
             "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    incs = ["-I" + p for p in ce.include_paths(device_type="cuda")] + ["-I" + sysconfig.get_paths()["include"]]
    libs = ["-L" + os.path.join(tdir, "lib"), "-L" + HERE, "-lwarp_rnnt_amd", "-lc10", "-lc10_hip", "-ltorch_cpu",
            "-ltorch_hip", "-ltorch", "-ltorch_python",
            "-Wl,-rpath,$ORIGIN/../warp_rnnt_amd", "-Wl,-rpath," + os.path.join(tdir, "lib")]
    fp = _fingerprint([BINDING_SRC, hdr], flags + incs + [torch.__version__])
    if not force and os.path.exists(BINDING) and _read(BINDING + ".fingerprint") == fp:
        return BINDING
    cmd = [gxx] + flags + incs + [BINDING_SRC, "-o", BINDING] + libs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(BINDING + ".fingerprint", "w") as f:
        f.write(fp)
    return BINDING


def ensure_built():
    """Build the library if it is missing, safely when several ranks of one node start together
    (exclusive file lock; the others find it built).  This is what bench.py and smoke() call; the
    package itself never builds on import and fails loudly without the library (_lib.load)."""
    if os.path.exists(LIB):
        return LIB
    import fcntl
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not os.path.exists(LIB):
                build()
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(verbose=True, variant=sys.argv[1] if len(sys.argv) > 1 else None))
