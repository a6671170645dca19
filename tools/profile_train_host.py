"""Host-side cProfile of the optimisation step of the 3D path (TrainStep): own time per function.
    python tools/profile_train_host.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eprecon_amd.fragment_step import TrainStep  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
s = TrainStep(seed=0, lr=1e-6)      # (lr: see bench.py, extra_workloads.train)
for _ in range(3):
    s.run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    s.run()
torch.cuda.synchronize()
print(f"train step {(time.perf_counter() - t0) / n * 1e3:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    s.run()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
rows = [(tt / n * 1e3, ct / n * 1e3, nc / n, f"{os.path.basename(fn)}:{line}:{name}") for (fn, line, name), (cc, nc, tt, ct, _) in st.stats.items()]
rows.sort(reverse=True)
print("# host: own ms/step | cumulative | calls | function")
for r in rows[:50]:
    print(f"{r[0]:8.3f} {r[1]:8.3f} {r[2]:8.1f}  {r[3]}")
if "--callers" in sys.argv:
    st.print_callers("_named_members|named_modules|zero_grad|_foreach")
rows.sort(key=lambda r: -r[1])
print("# by cumulative time")
for r in rows[:60]:
    print(f"{r[0]:8.3f} {r[1]:8.3f} {r[2]:8.1f}  {r[3]}")
