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
    from eprecon_amd import _lib
    data = open(_lib.LIB_PATH, "rb").read()
    archs = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", data))
    assert archs == {b"gfx950"}, archs


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
                                            ("eprecon_mlp4x_head", "Mlp4xHead"), ("eprecon_mlp4x_desc", "Mlp4xDesc")])
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
