"""The 2D feeder alone (MnasMulti.forward_views: 9 views of 640x480 as one channels-last batch, train-mode BatchNorm per view):
wall time per call and, under rocprofv3 --kernel-trace --stats, its kernels.
    python tools/profile_backbone.py [calls]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eprecon_amd.backbone import MnasMulti  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    torch.manual_seed(0)
    net = MnasMulti(1.0).cuda().train()
    imgs = [torch.randn(1, 3, 480, 640, device="cuda") for _ in range(9)]
    for _ in range(5):
        net.forward_views(imgs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        net.forward_views(imgs)
    torch.cuda.synchronize()
    print(f"forward_views: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per call (9 x 640x480)")


if __name__ == "__main__":
    with torch.no_grad():
        main()
