"""Timing-only ablations of the 16x16x4 MFMA kernels and the cross-workgroup split-K kernel (EPRECON_D3_ABLATE bits the current
kernels honour: 1 no MFMA loop (prologue + epilogue only); 2 — 16-row tile kernel only — every weight load from the first
offset's address (L1 hits)).  The first version of the direct gather kernel also had bits 8 (gathers from the row itself) and 16
(gathers sent out of range): those measurements are in profiles/r03/conv_direct_ablate.txt.
    python tools/conv_ablate.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eprecon_amd import sparse as SP  # noqa: E402
from eprecon_amd.fragment_step import Cfg2Step  # noqa: E402
from conv_shapes_ab import coords_for  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda")
    rng = np.random.default_rng(0)
    cases = []
    step = Cfg2Step(seed=0)
    coords = step.run()["init"][1].contiguous()
    vs = SP.VoxelSet(coords, 2, dims=(48, 48, 48))
    cases.append(("tile16 94k 32->32", SP.DenseMap(vs, (48, 48, 48)), coords.shape[0], 32, 32, (0, 2, 1)))
    for n, ci, co in ((320868, 48, 24), (93513, 48, 48), (198184, 24, 24), (57444, 96, 48)):
        v = SP.VoxelSet(torch.from_numpy(coords_for(n, rng)).to(dev), 1)
        cases.append((f"direct16 {n} {ci}->{co}", v.kernel_map(3), n, ci, co, (0, 1)))
    for n, ci, co in ((9415, 192, 96), (11880, 160, 80)):
        v = SP.VoxelSet(torch.from_numpy(coords_for(n, rng)).to(dev), 1)
        cases.append((f"wide {n} {ci}->{co}", v.kernel_map(3), n, ci, co, (0, 1)))
    with torch.no_grad():
        for name, m, n, ci, co, abls in cases:
            x = torch.randn(n, ci, device=dev)
            w = torch.randn(27, ci, co, device=dev) * 0.05
            b = torch.zeros(co, device=dev)
            out = torch.empty(n, co, device=dev)
            line = [name]
            for a in abls:
                os.environ["EPRECON_D3_ABLATE"] = str(a)
                line.append(f"abl{a}: {timed(lambda: SP.sparse_conv(x, w, m, b, out=out)):6.1f} us")
            os.environ["EPRECON_D3_ABLATE"] = "0"
            print(" | ".join(line))


if __name__ == "__main__":
    main()
