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
# --offload-compress: the gfx950 code objects are stored zstd-compressed in the fat binary (7.6 -> 2.x MB; rocPRIM's radix sort
# alone is 2.3 MB uncompressed) and unpacked by the HIP runtime at load
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "--offload-compress"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    return sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")] + [
        os.path.join(HERE, "..", "include", "eprecon_hip.h")]


def build(force=False, verbose=False):
    if not force and os.path.exists(OUT) and all(
            os.path.getmtime(d) <= os.path.getmtime(OUT) for d in _deps()):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, jobs = [], []
    for src in sources():
        obj = os.path.join(CSRC, os.path.basename(src)[:-4] + ".o")
        if force or not os.path.exists(obj) or any(
                os.path.getmtime(d) > os.path.getmtime(obj)
                for d in [src] + [x for x in _deps() if x.endswith((".hpp", ".h"))]):
            cmd = [hipcc] + [f for f in FLAGS if f != "-shared"] + ["-c", src, "-o", obj]
            if verbose:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
                print(" ".join(cmd))
            jobs.append(cmd)
        objs.append(obj)
    # the translation units are independent: compile them side by side
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1) or 1) as pool:
        list(pool.map(subprocess.check_call, jobs))
    cmd = [hipcc, "--offload-arch=gfx950", "--offload-compress", "-shared", "-fPIC"] + objs + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
