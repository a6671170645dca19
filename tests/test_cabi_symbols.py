"""CPU-only: the C-ABI library builds, loads, and exports every symbol include/eprecon_hip.h
declares; the ctypes table in eprecon_amd/_lib.py covers the same set.  No compute calls."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "eprecon_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(eprecon_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from eprecon_amd import build, _lib
    build.build()
    return _lib.load()


def test_header_declares_something():
    syms = declared_symbols()
    assert "eprecon_back_project_async" in syms and len(syms) >= 5


def test_library_exports_every_declared_symbol(lib):
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/eprecon_hip.h but not exported"


def test_ctypes_table_matches_header():
    from eprecon_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_build_identity(lib):
    assert lib.eprecon_abi_version() == 1
    assert lib.eprecon_build_arch() == b"gfx950"


def test_only_gfx950_code_objects():
    """the fat binary must carry gfx950 code only (no multi-arch / compatibility builds)"""
    import struct
    import subprocess
    import tempfile
    from eprecon_amd import _lib
    data = open(_lib.LIB_PATH, "rb").read()
    archs = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", data))
    # the code objects are stored compressed (--offload-compress: "CCOB" bundles, one per translation unit): each is handed to
    # clang-offload-bundler, which lists the targets it holds
    bundler = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
    n_bundles = 0
    for m in re.finditer(b"CCOB", data):
        o = m.start()
        ver, method = struct.unpack_from("<HH", data, o + 4)
        if ver not in (2, 3) or method > 1:
            continue                                  # (the four bytes somewhere else in the file)
        total = struct.unpack_from("<Q", data, o + 8)[0] if ver >= 3 else struct.unpack_from("<I", data, o + 8)[0]
        with tempfile.NamedTemporaryFile(suffix=".bin") as f:
            f.write(data[o:o + total])
            f.flush()
            out = subprocess.run([bundler, "--list", "--type=o", f"--input={f.name}"], capture_output=True, text=True)
        if out.returncode == 0:
            n_bundles += 1
            archs.update(t.rsplit("--", 1)[1].encode() for t in out.stdout.split() if "amdgcn" in t)
    assert archs == {b"gfx950"}, archs
    assert n_bundles == 0 or n_bundles >= 10          # compressed build: every translation unit was inspected


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from eprecon_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.EpreconError):
        _lib.load()


def test_operators_refuse_cpu_tensors():
    import torch
    from eprecon_amd import _lib
    from eprecon_amd.back_project import Back_Project
    coords = torch.zeros((4, 4), dtype=torch.int32)
    feats = torch.zeros((9, 1, 24, 8, 8))
    with pytest.raises(_lib.EpreconError):
        Back_Project(24)(coords, torch.zeros(1, 3), 0.04, feats, torch.eye(4).expand(9, 1, 4, 4), 0)


@pytest.mark.parametrize("c_name,py_name", [("eprecon_conv_desc", "ConvDesc"), ("eprecon_gru_stage_desc", "GruStageDesc"),
                                            ("eprecon_decoder_layer_desc", "DecoderLayerDesc"),
                                            ("eprecon_mlp4x_head", "Mlp4xHead"), ("eprecon_mlp4x_desc", "Mlp4xDesc"),
                                            ("eprecon_gru_finish_desc", "GruFinishDesc"), ("eprecon_spvcnn_geometry_desc", "SpvcnnGeometryDesc"),
                                            ("eprecon_spvcnn_conv", "SpvcnnConv"), ("eprecon_spvcnn_bn", "SpvcnnBn"),
                                            ("eprecon_spvcnn_forward_desc", "SpvcnnForwardDesc")])
def test_struct_layouts_match_header(tmp_path, c_name, py_name):
    """the ctypes mirrors of the descriptor structs have the size and field offsets the C compiler gives the header's
    structs (a drift here would silently corrupt every launch that goes through them)"""
    import ctypes
    import subprocess
    from eprecon_amd import _lib
    mirror = getattr(_lib, py_name)
    fields = [f[0] for f in mirror._fields_]
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "eprecon_hip.h"', 'int main(void) {',
           f'  printf("%zu\\n", sizeof({c_name}));']
    src += [f'  printf("%zu\\n", offsetof({c_name}, {name}));' for name in fields]
    src += ['  return 0;', '}']
    c_file = tmp_path / "layout.c"
    c_file.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c_file), "-o", str(exe)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert out[0] == ctypes.sizeof(mirror)
    assert out[1:] == [getattr(mirror, name).offset for name in fields]
