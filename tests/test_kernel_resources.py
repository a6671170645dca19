"""Static properties of the built gfx950 code objects (no GPU): read from the AMDHSA metadata by tools/kernel_resources.py.
A kernel of this repository that starts to spill (scratch > 0), or a hot kernel whose register count drops its occupancy, is a
performance regression no parity test sees."""
import os
import sys
import tempfile

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.fixture(scope="module")
def kernels():
    import kernel_resources as KR
    from eprecon_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    rows = []
    with tempfile.TemporaryDirectory() as wd:
        for elf in KR.code_objects(_lib.LIB_PATH, wd):
            rows += KR.kernels_of(elf)
    names = KR.demangle([r["name"] for r in rows])
    return [dict(r, demangled=n) for r, n in zip(rows, names)]


def test_every_translation_unit_was_read(kernels):
    assert len(kernels) > 200
    assert {int(k["wavefront_size"]) for k in kernels} == {64}


def test_no_kernel_of_this_repository_uses_scratch(kernels):
    ours = [k for k in kernels if "rocprim::" not in k["demangled"]]
    # (SGPR spills go to VGPR lanes and VGPR spills to free accumulator registers first: neither touches memory; scratch does)
    spilling = [(k["demangled"][:80], k["private_segment_fixed_size"]) for k in ours if int(k["private_segment_fixed_size"])]
    assert not spilling, spilling


@pytest.mark.parametrize("needle,min_waves", [
    ("spconv_direct16_kernel<2, 3, false, 3, false, 0>", 3),     # the round-5 cfg4 leader: 144 unified VGPRs
    ("spconv_direct16_kernel<2, 3, true, 3, false, 0>", 2),      # ... and its 16 + 8 form (round 6: eight more B registers per chunk)
    ("spconv_direct16_kernel<2, 3, true, 3, false, 1>", 3),      # ... with the loads spread among the MFMAs (the cfg4 leader now): 102 + 40
    ("spconv_direct16_kernel<3, 6, false, 3, false, 1>", 3),     # 96 -> 48 likewise
    ("conv3d_tile16_kernel<2, 2>", 7),          # the cfg2 sparse stack on dense grids
    ("bp_gather_mlp_kernel<64, 2, 8, 1>", 8),   # the kernel the bench line names
])
def test_hot_kernels_keep_their_occupancy(kernels, needle, min_waves):
    hit = [k for k in kernels if needle in k["demangled"]]
    assert hit, needle
    for k in hit:
        unified = (int(k["vgpr_count"]) + 7) // 8 * 8
        assert min(8, 512 // unified) >= min_waves, (k["demangled"][:80], k["vgpr_count"])
