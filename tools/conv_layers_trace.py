"""Each 2D-stack layer 10x through the HIP conv + BatchNorm-summary path (run under rocprofv3 --kernel-trace;
summarise with tools/conv_layers_summary.py): kernel durations without host overhead or concurrency."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eprecon_amd import dense2d as D2

dev = torch.device("cuda:0")
V = 9
layers = []
for (c, h, w) in ((80, 30, 40), (40, 60, 80), (24, 120, 160)):
    hh = c // 2
    layers += [(h, w, c, c, 3), (h, w, c, c, 1), (h, w, c, hh, 3), (h, w, hh, hh, 3), (h, w, 4 * c, c, 1)]
layers += [(60, 80, 144, 32, 1), (60, 80, 32, 32, 3)]
torch.manual_seed(0)
with torch.no_grad():
    for (h, w, ci, co, k) in layers:
        x = D2.Act(torch.randn(V * h * w, ci, device=dev))
        wk = torch.randn(k * k, ci, co, device=dev) * 0.1
        b = torch.randn(co, device=dev)
        g = torch.ones(co, device=dev)
        grid = D2.PixelGrid.get(V, h, w, dev)
        out = torch.empty(V * h * w, co, device=dev)
        torch.cuda.synchronize()
        for _ in range(10):
            D2.conv_bn_launch(wk, b, g, b, 1e-5, k, x, grid, out=out)
        torch.cuda.synchronize()
        print(f"{h}x{w} {ci}->{co} k{k}")
