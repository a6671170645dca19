"""Where the drop-in boundary spends a fragment: NeuralRecon.forward split into the 2D backbones, NeuConNet.forward and
fuse_to_global, with a device synchronisation around each (so the parts add up to more than the free-running call).
    python tools/profile_e2e.py [n_rounds]"""
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eprecon_amd.fragment_step import E2EStep  # noqa: E402

acc = collections.OrderedDict()


def timed(name, fn):
    def wrap(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0)
        return r
    return wrap


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    step = E2EStep(seed=0, device=torch.device("cuda"))
    n = rounds * step.n_fragments
    for _ in range(step.n_fragments):
        step.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step.run()
    torch.cuda.synchronize()
    free = (time.perf_counter() - t0) / n * 1e3
    m = step.model
    m.backbone2d.forward_views = timed("backbone2d (9 views, one batch)", m.backbone2d.forward_views)
    m.backbone_occ_pano.forward_views = timed("backbone_occ_pano", m.backbone_occ_pano.forward_views)
    m.neucon_net.forward = timed("NeuConNet.forward", m.neucon_net.forward)
    m.fuse_to_global.forward = timed("fuse_to_global", m.fuse_to_global.forward)
    step.voxels.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step.run()
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / n * 1e3
    print(f"free-running {free:.2f} ms/fragment; with per-stage syncs {total:.2f} ms/fragment; finest voxels {step.voxels}")
    for k, v in acc.items():
        print(f"  {k:34s} {v / n * 1e3:7.3f} ms")
    print(f"  {'rest (normalisation, glue)':34s} {total - sum(acc.values()) / n * 1e3:7.3f} ms")


if __name__ == "__main__":
    with torch.no_grad():
        main()
