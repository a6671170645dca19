"""A/B of the back-projection variants on the dense 96^3 level (env toggles read once per process)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eprecon_amd import _lib, back_project as BP
from eprecon_amd.fragment_step import Cfg2Step
lvl = int(os.environ.get("LVL", "0"))
step = Cfg2Step(seed=0)
lib = _lib.load()
interval = {0: 1, 1: 2, 2: 4}[lvl]
run = lambda: BP.run(step.coords[interval], step.origin, 0.04, step.feats[lvl], step.krcam[lvl], 0)
for _ in range(5): run()
lib.eprecon_profile_enable(1)
g = []
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30):
    run(); g.append(lib.eprecon_profile_gather_ms())
torch.cuda.synchronize()
print(f"mlp={os.environ.get('EPRECON_BP_MLP','1')} lvl={lvl}: "
      f"op {(time.perf_counter()-t0)/30*1e3:.3f} ms, gather kernel {sum(g)/len(g)*1e3:.1f} us")
