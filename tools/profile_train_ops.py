"""torch.profiler view of one optimisation step of the 3D path: operators / autograd nodes by host time
    python tools/profile_train_ops.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eprecon_amd.fragment_step import TrainStep  # noqa: E402

s = TrainStep(seed=0, lr=1e-6)
for _ in range(3):
    s.run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(2):
        s.run()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cpu_time_total", row_limit=70, max_name_column_width=60))
