"""Point-wise layers (K = 1, identity map) of a cfg4 fragment that run on the slab kernel (C_out > 64 or medium lists): time per
launch from HIP events, alone on the device, with the BatchNorm summaries they produce.   python tools/conv_k1_shapes.py [tag]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eprecon_amd import _lib  # noqa: E402
from eprecon_amd import sparse as SP  # noqa: E402

SHAPES = [(74568, 96, 96), (74568, 80, 80), (93512, 48, 96), (11880, 192, 192), (11880, 192, 96), (11880, 160, 160), (11880, 160, 80),
          (172800, 96, 24), (43200, 160, 40), (43200, 144, 32), (10800, 80, 80), (11744, 128, 96), (9324, 128, 96), (7561, 80, 48),
          (1532, 160, 96), (11880, 96, 48), (9324, 64, 32), (7561, 48, 48), (4000, 64, 64), (320868, 24, 24), (198184, 32, 24), (57444, 96, 48)]
tag = sys.argv[1] if len(sys.argv) > 1 else "default"
if len(sys.argv) > 2:      # rows from which point-wise layers with C_out <= 64 take the direct kernel (sparse.K1_DIRECT_MIN_ROWS)
    SP.K1_DIRECT_MIN_ROWS = int(sys.argv[2])
print(f"# {tag}: rows C_in -> C_out | us per launch | kernel | GB/s of (rows x (C_in + C_out) x 4 B) | TF")
tot = 0.0
with torch.no_grad():
    for n, ci, co in SHAPES:
        x = torch.randn(n, ci, device="cuda")
        w = torch.randn(1, ci, co, device="cuda") * 0.05
        out = torch.empty(n, co, device="cuda")
        run = lambda: SP.conv_stats(x, w, None, out=out)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        tot += us
        print(f"{n:7d} {ci:4d} -> {co:3d} | {us:7.1f} | {_lib.last_conv_kernel():24s} | {n * (ci + co) * 4 / us / 1e3:7.0f} | {2.0 * n * ci * co / us / 1e6:5.1f}")
print(f"sum {tot:.0f} us")
