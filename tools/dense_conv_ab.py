"""A/B: the 2D fusion-stack convolutions through MIOpen (channels-last F.conv2d) vs the HIP gather-GEMM
kernel driven by a dense 2D neighbour table (a 3x3 'same' convolution over [V,H,W] pixels is a sparse
convolution whose kernel map is known in closed form)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from eprecon_amd import sparse as SP


def dense_table(v, h, w, dev):
    idx = torch.arange(v * h * w, device=dev, dtype=torch.int32).view(v, h, w)
    tab = []
    for ky in (-1, 0, 1):
        for kx in (-1, 0, 1):
            t = torch.full((v, h, w), -1, dtype=torch.int32, device=dev)
            ys, ye = max(0, -ky), min(h, h - ky)
            xs, xe = max(0, -kx), min(w, w - kx)
            t[:, ys:ye, xs:xe] = idx[:, ys + ky:ye + ky, xs + kx:xe + kx]
            tab.append(t.reshape(-1))
    return torch.stack(tab).contiguous()


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


dev = torch.device("cuda:0")
V = 9
layers = []
for (c, h, w) in ((80, 30, 40), (40, 60, 80), (24, 120, 160)):
    hh = c // 2
    layers += [(h, w, c, c, 3), (h, w, c, c, 1), (h, w, c, hh, 3), (h, w, hh, hh, 3), (h, w, 4 * c, c, 1)]
layers += [(60, 80, 144, 32, 1), (60, 80, 32, 32, 3)]
torch.manual_seed(0)
tot_a = tot_b = 0.0
mult = {0: 1, 1: 3, 2: 1, 3: 3, 4: 1}
with torch.no_grad():
    for li, (h, w, ci, co, k) in enumerate(layers):
        x = torch.randn(V, ci, h, w, device=dev).contiguous(memory_format=torch.channels_last)
        wt = torch.randn(co, ci, k, k, device=dev) * 0.1
        wt_cl = wt.contiguous(memory_format=torch.channels_last)
        b = torch.randn(co, device=dev)
        rows = x.permute(0, 2, 3, 1).reshape(V * h * w, ci)
        wk = wt.permute(2, 3, 1, 0).reshape(k * k, ci, co).contiguous()
        tab = dense_table(V, h, w, dev) if k == 3 else None
        ref = F.conv2d(x, wt_cl, b, padding="same")
        out = SP.sparse_conv(rows, wk if k == 3 else wk[0], tab, b)
        err = (out - ref.permute(0, 2, 3, 1).reshape(V * h * w, co)).abs().max().item()
        ta = timeit(lambda: F.conv2d(x, wt_cl, b, padding="same"))
        tb = timeit(lambda: SP.sparse_conv(rows, wk if k == 3 else wk[0], tab, b))
        gf = 2.0 * V * h * w * ci * co * k * k / 1e9
        print(f"{h}x{w} {ci}->{co} k{k}: miopen+bias {ta:7.1f} us  hip {tb:7.1f} us  ({gf / tb * 1e3:5.1f} TF)  max|d| {err:.2e}")
        if li < 15:
            m = mult[li % 5] if li % 5 != 1 else 3  # 1x1 C->C appears three times per Fusion_Block
            tot_a += ta * m; tot_b += tb * m
        elif li == 15:
            tot_a += ta; tot_b += tb
        else:
            tot_a += 4 * ta; tot_b += 4 * tb
print(f"stack total (host-timed, back-to-back launches): miopen+bias {tot_a / 1e3:.3f} ms   hip {tot_b / 1e3:.3f} ms")
