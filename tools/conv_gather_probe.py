"""Where does the gather-GEMM lose its time?  The same 27-offset convolution (dense-grid 48^3 list, 32 -> 32 and
48 -> 24 / 96 -> 48) with kernel maps of different locality but the SAME number of live pairs:
  real      the 3x3x3 neighbourhood of the list
  self      every offset points at the row itself (every gather after the first hits the L1 line just loaded)
  shift     offset k reads row i + k (consecutive rows: perfectly coalesced, L2-resident)
  random    offset k reads a random row (no locality at all)
If `self` is much faster than `real`, the kernel is bound by the gathers (L2 latency / bandwidth), not by MFMA issue."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eprecon_amd import sparse as SP  # noqa: E402
from eprecon_amd import synthetic as S  # noqa: E402


def timed(x, w, nbr, out, reps=20):
    for _ in range(3):
        SP.conv_stats(x, w, nbr, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        SP.conv_stats(x, w, nbr, out=out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda")
    rng = np.random.default_rng(0)
    coords = S.dense_coords((96, 96, 96), 2)
    keep = np.sort(rng.choice(len(coords), 94000, replace=False))
    vs = SP.VoxelSet(torch.from_numpy(np.ascontiguousarray(coords[keep])).to(dev), 2)
    real = vs.kernel_map(3)
    n = real.shape[1]
    live = real >= 0
    rows = torch.arange(n, device=dev, dtype=torch.int32)
    maps = {"real": real,
            "self": torch.where(live, rows[None, :].expand(27, n), real).contiguous(),
            "shift": torch.where(live, ((rows[None, :] + torch.arange(27, device=dev, dtype=torch.int32)[:, None]) % n), real).contiguous(),
            "random": torch.where(live, torch.randint(0, n, (27, n), device=dev, dtype=torch.int32), real).contiguous()}
    with torch.no_grad():
        for ci, co in ((32, 32), (48, 24), (96, 48), (64, 64)):
            x = torch.randn(n, ci, device=dev)
            w = torch.randn(27, ci, co, device=dev) * 0.05
            out = torch.empty(n, co, device=dev)
            pairs = int(live.sum())
            line = f"{ci:3d}->{co:3d} pairs {pairs}: "
            for name, nbr in maps.items():
                us = timed(x, w, nbr, out)
                line += f" {name} {us:7.1f} us ({2.0 * pairs * ci * co / us / 1e6:5.1f} TF)"
            print(line)


main()
