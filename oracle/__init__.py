"""CPU oracle for the per-fragment 3D hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and
only as the checker / the timed CPU baseline.  The product (eprecon_amd/) never imports it and
has no CPU fallback: it raises when the HIP library is missing.

Parity pins (see DESIGN.md "Oracle"):
  * pieces the reference implements in importable Python (back-projection, grids, upsample,
    morphology, GRU-fusion bookkeeping, dense heads) are pinned against golden vectors captured
    from the reference itself by tests/golden/make_golden.py;
  * pieces whose arithmetic lives in the un-vendored torchsparse / spconv CUDA extensions
    (README.md:19, requirements.txt:18 of the reference) are restated from their published
    semantics and the reference's call sites -> "parity unpinned" for those.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, "c", f) for f in sorted(os.listdir(os.path.join(_HERE, "c")))
            if f.endswith(".c")]
    stale = force or not os.path.exists(so) or any(
        os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB
