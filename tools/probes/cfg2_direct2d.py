"""cfg2 step with the 3x3 layers of the 43,200-pixel level on the direct gather kernel (DIRECT_2D_MIN_ROWS lowered) against the
image-tile kernel: python tools/probes/cfg2_direct2d.py [min_rows]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eprecon_amd import dense2d as D2  # noqa: E402
from eprecon_amd.fragment_step import Cfg2Step  # noqa: E402

if len(sys.argv) > 1:
    D2.DIRECT_2D_MIN_ROWS = int(sys.argv[1])
if len(sys.argv) > 2:
    D2.K1_DIRECT_2D_MIN_ROWS = int(sys.argv[2])
step = Cfg2Step(seed=0, device=torch.device("cuda"))
step.defer_reads = True
for _ in range(30):
    step.run()
step.flush()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    step.run()
step.flush()
torch.cuda.synchronize()
print(f"DIRECT_2D_MIN_ROWS={D2.DIRECT_2D_MIN_ROWS} K1={D2.K1_DIRECT_2D_MIN_ROWS}: {(time.perf_counter() - t0) / 300 * 1e3:.4f} ms per step")
