"""Builds eprecon_amd/libeprecon_hip.so (gfx950 only) with hipcc, in-tree.

    python -m eprecon_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels with the repo
snapshot to the GPU box.  -ffp-contract=off is part of the arithmetic contract (csrc/common.hpp).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libeprecon_hip.so")
# --offload-compress: the gfx950 code objects are stored zstd-compressed in the fat binary (5.3 -> 1.4 MB since round 5, when the
# last library kernel — rocPRIM's radix sort, 2.3 MB uncompressed — left csrc/hash_order.hip) and unpacked by the HIP runtime at load
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "--offload-compress"]


# Per-file flags.  sparse_conv_direct.hip: `-fno-honor-nans` — with a pending BatchNorm + ReLU on the input the direct gather
# kernel applies fmaxf(x, 0) to every gathered value, and clang puts a canonicalising `v_max_f32 x, x, x` in front of each one
# (llvm.maxnum must quiet signalling NaNs): 8 extra VALU instructions per gathered quad next to its 8 MFMAs.  The flag removes
# exactly those (checked on the assembly, DESIGN.md section 8: 3,472 -> 2,576 v_max_f32 over the file, nothing else changes);
# v_max_f32 itself still returns the non-NaN operand, so ReLU(NaN) = 0 as before.  Nothing in that file feeds an index decision.
EXTRA_FLAGS = {"sparse_conv_direct.hip": ["-fno-honor-nans"]}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    return sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")] + [
        os.path.join(HERE, "..", "include", "eprecon_hip.h")]


def build(force=False, verbose=False, out=OUT, extra_flags=None, obj_suffix=".o"):
    """out / extra_flags / obj_suffix: an A/B variant of the library next to the shipped one (tools/: e.g. the build WITHOUT
    the per-file flags above as libeprecon_hip_plain.so, loaded through EPRECON_LIB_PATH by the timing tools only)"""
    extra_flags = EXTRA_FLAGS if extra_flags is None else extra_flags
    stamp = os.path.abspath(__file__)      # (the flags live in this file: a change here rebuilds)
    if not force and os.path.exists(out) and all(
            os.path.getmtime(d) <= os.path.getmtime(out) for d in _deps() + [stamp]):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, jobs = [], []
    for src in sources():
        obj = os.path.join(CSRC, os.path.basename(src)[:-4] + obj_suffix)
        if force or not os.path.exists(obj) or any(
                os.path.getmtime(d) > os.path.getmtime(obj)
                for d in [src, stamp] + [x for x in _deps() if x.endswith((".hpp", ".h"))]):
            cmd = [hipcc] + [f for f in FLAGS if f != "-shared"] + extra_flags.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
            if verbose:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
                print(" ".join(cmd))
            jobs.append(cmd)
        objs.append(obj)
    # the translation units are independent: compile them side by side
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1) or 1) as pool:
        list(pool.map(subprocess.check_call, jobs))
    cmd = [hipcc, "--offload-arch=gfx950", "--offload-compress", "-shared", "-fPIC"] + objs + ["-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if "--plain" in sys.argv:    # the A/B twin without the per-file flags (not loaded by the package unless EPRECON_LIB_PATH says so)
        print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, out=os.path.join(HERE, "libeprecon_hip_plain.so"),
                    extra_flags={}, obj_suffix=".plain.o"))
    else:
        print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
