"""Wall time of one optimisation step of the 3D path (TrainStep) and, under rocprofv3 --kernel-trace --stats, its
kernel breakdown.  python tools/profile_train_step.py [steps]"""
import sys
import time

import torch

from eprecon_amd.fragment_step import TrainStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
s = TrainStep(seed=0, lr=1e-6)      # (lr: see bench.py, extra_workloads.train)
for _ in range(3):
    s.run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    last = s.run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
with torch.no_grad():
    s.net.gru_fusion.scene_name = [None, None, None]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        s.net.gru_fusion.scene_name = [None, None, None]
        s.net(s.f1, s.f2, s.inputs, {})
    torch.cuda.synchronize()
    inf = (time.perf_counter() - t0) / steps
print(f"train step {dt * 1e3:.2f} ms   inference forward (same fragment, scene restarted) {inf * 1e3:.2f} ms   loss {last['total_loss']:.4f}")
