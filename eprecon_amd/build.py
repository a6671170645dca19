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


# Per-file flags (none since round 6: the direct gather kernel's ReLU-on-load is a v_med3_f32 now — csrc/sparse_conv_direct_impl.hpp
# `relu_nc` — instead of a translation-unit-wide -fno-honor-nans).
EXTRA_FLAGS = {}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    return sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")] + [
        os.path.join(HERE, "..", "include", "eprecon_hip.h")]


def build(force=False, verbose=False, out=OUT, extra_flags=None, obj_dir=CSRC, defines=(), only=None):
    """out / obj_dir / extra_flags / defines: an A/B variant of the library built AWAY from the package directory (the timing
    tools build it inside their own run, e.g. under gpurun_out/variants/, and load it through EPRECON_LIB_PATH; nothing but
    libeprecon_hip.so ever sits beside the package: VERDICT r05 housekeeping)"""
    extra_flags = EXTRA_FLAGS if extra_flags is None else extra_flags
    os.makedirs(obj_dir, exist_ok=True)
    stamp = os.path.abspath(__file__)      # (the flags live in this file: a change here rebuilds)
    if not force and os.path.exists(out) and all(
            os.path.getmtime(d) <= os.path.getmtime(out) for d in _deps() + [stamp]):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, jobs = [], []
    for src in sources():
        # (a name selects its family: sparse_conv_direct.hip also means sparse_conv_direct_ct1.hip ... — the kernels of
        # csrc/sparse_conv_direct_impl.hpp are instantiated by those)
        if only is not None and not any(os.path.basename(src)[:-4].startswith(o[:-4]) for o in only):    # a variant of a few translation units: the rest are the
            objs.append(os.path.join(CSRC, os.path.basename(src)[:-4] + ".o"))    # main build's objects (built first)
            continue
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        if force or not os.path.exists(obj) or any(
                os.path.getmtime(d) > os.path.getmtime(obj)
                for d in [src, stamp] + [x for x in _deps() if x.endswith((".hpp", ".h"))]):
            cmd = ([hipcc] + [f for f in FLAGS if f != "-shared"] + list(defines) + extra_flags.get(os.path.basename(src), [])
                   + ["-c", src, "-o", obj])
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


def build_variant(name, defines=(), extra_flags=None, root=None, only=None):
    """an A/B twin under <root>/variants/<name>/ (default root: gpurun_out, which never travels and is git-ignored); only = the
    translation units the defines concern (the others are linked from the main build)"""
    root = root or os.environ.get("EPRECON_VARIANT_ROOT") or os.path.join(HERE, "..", "gpurun_out")
    d = os.path.abspath(os.path.join(root, "variants", name))
    if only is not None:
        build()
    return build(out=os.path.join(d, "libeprecon_hip.so"), obj_dir=d, defines=defines, extra_flags=extra_flags, only=only)


if __name__ == "__main__":
    if "--variant" in sys.argv:    # python -m eprecon_amd.build --variant NAME [-DX=1 ...]: see build_variant
        only = [a for a in sys.argv if a.endswith(".hip")] or None     # e.g. sparse_conv_direct.hip
        print(build_variant(sys.argv[sys.argv.index("--variant") + 1], defines=[a for a in sys.argv if a.startswith("-D")], only=only))
    else:
        print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
