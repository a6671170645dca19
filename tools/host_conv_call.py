"""host-side cost of one convolution call through the Python wrappers (tiny list: the GPU is never the limit)"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eprecon_amd import sparse as SP  # noqa: E402

dev = torch.device("cuda")
rng = np.random.default_rng(0)
c = np.unique(rng.integers(0, 12, (600, 3)), axis=0)
coords = np.concatenate([np.zeros((len(c), 1), np.int64), c], 1).astype(np.int32)
vs = SP.VoxelSet(torch.from_numpy(coords).to(dev), 1)
nbr = vs.kernel_map(3)
x = torch.randn(vs.n, 32, device=dev)
w = torch.randn(27, 32, 32, device=dev)
out = torch.empty(vs.n, 32, device=dev)
ln = torch.nn.LayerNorm(32).to(dev)
b = torch.zeros(32, device=dev)
calls = {"conv_stats": lambda: SP.conv_stats(x, w, nbr, out=out),
         "sparse_conv": lambda: SP.sparse_conv(x, w, nbr, None, out=out),
         "sparse_conv_ln": lambda: SP.sparse_conv_ln(x, w, nbr, b, ln.weight, ln.bias, ln.eps, out=out)}
with torch.no_grad():
    for name, fn in calls.items():
        for _ in range(200):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3000):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"{name:16s} {(t1 - t0) / 3000 * 1e6:6.1f} us of host time per call")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3000):
        calls["conv_stats"]()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(12)
