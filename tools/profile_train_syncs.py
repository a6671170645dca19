"""Where the optimisation step blocks the host: every synchronising call of one TrainStep.run (torch.cuda.set_sync_debug_mode:
nonzero, item, boolean-mask indexing, .cpu()), attributed to the Python line that issued it; calls made inside the autograd
engine (backward nodes) are attributed to the line that called backward.
    python tools/profile_train_syncs.py"""
import collections
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eprecon_amd.fragment_step import TrainStep  # noqa: E402

s = TrainStep(seed=0, lr=1e-6)
for _ in range(3):
    s.run()
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as rec:
    warnings.simplefilter("always")
    s.run()
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
sites = collections.Counter()
for w in rec:
    if "synchroniz" in str(w.message).lower():
        sites[f"{os.path.relpath(w.filename)}:{w.lineno}"] += 1
print(f"# {sum(sites.values())} synchronising calls in one step; calls | site")
for site, n in sites.most_common():
    print(f"{n:5d}  {site}")
