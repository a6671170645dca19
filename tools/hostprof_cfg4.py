"""Host-side profile of the cfg4 fragment loop (cProfile): where the Python thread spends its time while the GPU
runs asynchronously.  If the host total is close to the wall time per fragment the path is host-bound.
    python tools/hostprof_cfg4.py [n_fragments]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eprecon_amd.fragment_step import Cfg4Step  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
with torch.no_grad():
    step = Cfg4Step(seed=0, device=torch.device("cuda"))
    for _ in range(8):
        step.run()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for _ in range(n):
        step.run()
    pr.disable()
    torch.cuda.synchronize()
    print(f"wall {1e3 * (time.perf_counter() - t0) / n:.2f} ms/fragment (with cProfile overhead)")
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(60)
