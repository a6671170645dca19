"""Static resources of every kernel in libeprecon_hip.so, read from the gfx950 code objects' AMDHSA metadata (no GPU needed):
unified VGPRs (`.vgpr_count`: architectural + accumulator registers, as allocated), the accumulator part, SGPRs, LDS bytes,
scratch bytes (spills or local arrays), workgroup size limit and the resulting waves per SIMD (512 unified VGPRs per lane on
gfx950, allocated in units of 8: waves = min(8, 512 // round_up(vgpr_count, 8))).

    python tools/kernel_resources.py [> profiles/rNN/kernel_resources.txt]

LDS is the STATIC part only (`.group_segment_fixed_size`): kernels that size their LDS at launch (the gather-GEMM convolutions,
the attention kernels) show 0 here.  s-spill / v-spill: SGPRs kept in VGPR lanes and VGPRs kept in free accumulator registers
(neither touches memory; scratch does).  rocPRIM's radix-sort instantiations are summarised in one line.

The code objects are stored compressed (--offload-compress, "CCOB" bundles, one per translation unit); each is unbundled with
clang-offload-bundler and its notes are read with llvm-readelf.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(lib_path, workdir):
    data = open(lib_path, "rb").read()
    out = []
    for m in re.finditer(b"CCOB", data):
        o = m.start()
        ver, method = struct.unpack_from("<HH", data, o + 4)
        if ver not in (2, 3) or method > 1:
            continue
        total = struct.unpack_from("<Q", data, o + 8)[0] if ver >= 3 else struct.unpack_from("<I", data, o + 8)[0]
        src = os.path.join(workdir, f"bundle{len(out)}.bin")
        open(src, "wb").write(data[o:o + total])
        lst = subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={src}"], capture_output=True, text=True)
        if lst.returncode != 0:
            continue
        for target in lst.stdout.split():
            if "amdgcn" not in target:
                continue
            dst = os.path.join(workdir, f"co{len(out)}.elf")
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={src}", f"--targets={target}",
                                f"--output={dst}"], capture_output=True, text=True)
            if r.returncode == 0:
                out.append(dst)
    return out


def kernels_of(elf):
    import yaml
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", elf], capture_output=True, text=True).stdout
    rows = []
    for doc in re.findall(r"^\s*---\n(.*?)^\s*\.\.\.\s*$", notes, flags=re.S | re.M):   # one YAML document per metadata note
        meta = yaml.safe_load(doc)
        rows += [{k.lstrip("."): v for k, v in kern.items()} for kern in (meta or {}).get("amdhsa.kernels", [])]
    return rows


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines()


def main():
    from eprecon_amd import _lib
    with tempfile.TemporaryDirectory() as wd:
        rows = []
        for elf in code_objects(_lib.LIB_PATH, wd):
            rows += kernels_of(elf)
    names = demangle([r["name"] for r in rows])
    table = []
    for r, n in zip(rows, names):
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*$", "", n)
        v, a = int(r.get("vgpr_count", 0)), int(r.get("agpr_count", 0))
        unified = (v + 7) // 8 * 8
        waves = min(8, 512 // unified) if unified else 8
        table.append((n, v, a, int(r.get("sgpr_count", 0)), int(r.get("group_segment_fixed_size", 0)),
                      int(r.get("private_segment_fixed_size", 0)), int(r.get("max_flat_workgroup_size", 0)), waves,
                      int(r.get("sgpr_spill_count", 0)), int(r.get("vgpr_spill_count", 0))))
    table.sort(key=lambda t: (-t[5], -t[1], t[0]))
    lib_rows = [t for t in table if t[0].startswith("rocprim::")]
    table = [t for t in table if not t[0].startswith("rocprim::")]
    print(f"{len(table)} kernels of this repository in {os.path.basename(_lib.LIB_PATH)} ({os.path.getsize(_lib.LIB_PATH)} bytes); "
          f"{sum(1 for t in table if t[5])} use scratch (spills or local arrays)")
    if lib_rows:
        print(f"+ {len(lib_rows)} rocPRIM instantiations (csrc/hash_order.hip's radix sort): unified VGPRs <= {max(t[1] for t in lib_rows)}, "
              f"scratch <= {max(t[5] for t in lib_rows)} B ({sum(1 for t in lib_rows if t[5])} kernels)")
    print(f"{'kernel':96s} {'vgpr':>5s} {'acc':>5s} {'sgpr':>5s} {'lds B':>7s} {'scratch B':>9s} {'max wg':>6s} {'waves/SIMD':>10s} {'s-spill':>7s} {'v-spill':>7s}")
    for t in table:
        print(f"{t[0][:96]:96s} {t[1]:5d} {t[2]:5d} {t[3]:5d} {t[4]:7d} {t[5]:9d} {t[6]:6d} {t[7]:10d} {t[8]:7d} {t[9]:7d}")


if __name__ == "__main__":
    main()
