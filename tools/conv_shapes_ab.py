"""Microbenchmark of the gather-GEMM convolution on the layer shapes of cfg4 (3x3x3, stride 1): kernel time per
launch from HIP events around 20 back-to-back launches.  Run under different EPRECON_CONV_* switches to A/B.
    python tools/conv_shapes_ab.py [tag]
EPRECON_AB_IN_AFFINE=1: every launch applies a pending BatchNorm + ReLU to its input while gathering (the second convolution of
a ResidualBlock, models/modules.py:46-72): the path the per-file `-fno-honor-nans` of eprecon_amd/build.py is about.
EPRECON_LIB_PATH=gpurun_out/variants/NAME/libeprecon_hip.so: an A/B twin of the library (python -m eprecon_amd.build --variant NAME -DX=1)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eprecon_amd import sparse as SP  # noqa: E402

SHAPES = [  # (rows, C_in, C_out, what): the 3x3x3 launches of one cfg4 fragment (profiles/r03/cfg4_layers.txt)
    (9415, 192, 96, "ConvGRU voxel s0"), (9415, 160, 80, "ConvGRU img s0"), (11880, 192, 96, "ConvGRU voxel s0 (convr)"),
    (57444, 96, 48, "ConvGRU voxel s1"), (57444, 80, 40, "ConvGRU img s1"), (74568, 96, 48, "ConvGRU voxel s1 (convr)"),
    (320868, 48, 24, "ConvGRU s2"),
    (9324, 80, 32, "SPVCNN0 stem"), (9324, 128, 96, "SPVCNN0 up2"), (9324, 96, 96, "SPVCNN0 up2 res"),
    (1532, 32, 64, "SPVCNN0 stage1"), (1532, 64, 64, "SPVCNN0 stage1 res"), (1532, 160, 96, "SPVCNN0 up1"),
    (1532, 96, 96, "SPVCNN0 up1 res"), (204, 64, 128, "SPVCNN0 stage2"), (204, 128, 128, "SPVCNN0 stage2 res"),
    (28864, 140, 16, "SPVCNN1 stem (138 padded)"), (28864, 64, 48, "SPVCNN1 up2"), (28864, 48, 48, "SPVCNN1 up2 res"),
    (7561, 32, 32, "SPVCNN1 stage1 res"), (7561, 80, 48, "SPVCNN1 up1"), (7561, 48, 48, "SPVCNN1 up1 res"),
    (1502, 64, 64, "SPVCNN1 stage2 res"),
    (198184, 76, 8, "SPVCNN2 stem (74 padded)"), (198184, 32, 24, "SPVCNN2 up2"), (198184, 24, 24, "SPVCNN2 up2 res"),
    (47864, 16, 16, "SPVCNN2 stage1 res"), (47864, 24, 24, "SPVCNN2 up1 res"), (10121, 32, 32, "SPVCNN2 stage2 res"),
    (93513, 48, 48, "mask features"),
]


def coords_for(n, rng, density=0.35):
    d = int(np.ceil((n / density) ** (1 / 3)))
    flat = np.sort(rng.choice(d ** 3, size=n, replace=False))
    xyz = np.stack(np.unravel_index(flat, (d, d, d)), 1)
    return np.concatenate([np.zeros((n, 1), np.int64), xyz], 1).astype(np.int32)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "default"
    rng = np.random.default_rng(0)
    dev = torch.device("cuda")
    print(f"# {tag}: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith(("EPRECON_CONV", "EPRECON_AB", "EPRECON_LIB"))))
    total = 0.0
    with torch.no_grad():
        for n, ci, co, what in SHAPES:
            # Every shape on freshly mapped device memory (EPRECON_AB_EMPTY_CACHE=0: on whatever blocks the caching allocator
            # hands back).  On some boxes of the pool one shape — 76 -> 8 on 198,184 rows, the 24th of the list — took 1.7-2.7 ms
            # instead of 0.16 in 8 of 12 processes when its buffers were carved out of blocks cached from earlier shapes, under
            # every kernel selection, and in 0 of 12 with this line (profiles/r06/conv_shapes_placement.txt); other boxes never
            # showed it.  A property of where the buffers sit, not of the launch: the list is about the launches.
            if os.environ.get("EPRECON_AB_EMPTY_CACHE", "1") == "1":
                x = w = out = vs = nbr = None
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
            vs = SP.VoxelSet(torch.from_numpy(coords_for(n, rng)).to(dev), 1)
            nbr = vs.kernel_map(3)
            pairs = int((nbr >= 0).sum())
            x = torch.randn(n, ci, device=dev)
            w = torch.randn(27, ci, co, device=dev) * 0.05
            out = torch.empty(n, co, device=dev)
            aff = None
            if os.environ.get("EPRECON_AB_IN_AFFINE", "0") == "1":
                aff = (torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1, True)
            for _ in range(3):
                SP.conv_stats(x, w, nbr, out=out, in_affine=aff)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                SP.conv_stats(x, w, nbr, out=out, in_affine=aff)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            tf = 2.0 * pairs * ci * co / (us * 1e-6) / 1e12
            total += us
            print(f"{what:28s} N={n:7d} {ci:4d}->{co:3d}  {us:8.1f} us  {tf:6.1f} TF (live pairs)")
    print(f"sum {total:.0f} us")


if __name__ == "__main__":
    main()
