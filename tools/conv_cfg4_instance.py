"""The convolution that leads the cfg4 profile, alone on the device: 3x3x3 48 -> 24 of the finest-level ConvGRU (convz / convq) on
the REAL kernel map of a steady-state cfg4 fragment (the union voxels of the fragment and the scene map, ~320k rows).  20
back-to-back launches between two HIP events; live (row, offset) pairs counted from the map.  Run under
rocprofv3 --kernel-trace: the launches with this grid size are the committed row `roofline_conv_cfg4` is recomputed from.
    python tools/conv_cfg4_instance.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eprecon_amd import _lib  # noqa: E402
from eprecon_amd import sparse as SP  # noqa: E402
from eprecon_amd.fragment_step import Cfg4Step  # noqa: E402

from eprecon_amd import torchsparse_utils as TU  # noqa: E402


def main():
    step = Cfg4Step(seed=0, device=torch.device("cuda"))
    for _ in range(2 * step.n_fragments - 1):
        step.run()
    step.run()                                  # the last fragment of the scene: the largest map state
    torch.cuda.synchronize()
    # the kernel maps of the fragment's voxelisations are in the cache the SConv3d layers read (the finest level's last)
    maps = [e.vset._k3 for e in TU._VOX_CACHE if e.vset._k3 is not None and e.vset._k3.shape[1] > 200000]
    cands = [(float((m >= 0).float().mean()), m) for m in maps]
    cands = [c for c in cands if c[0] > 0.2]    # (the second ConvGRU voxelisation has no adjacent voxels: 1 / 27 live)
    live_frac, nbr = max(cands, key=lambda c: c[1].shape[1])
    n = nbr.shape[1]
    pairs = int((nbr >= 0).sum())
    x = torch.randn((n, 48), device="cuda")
    w = torch.randn((27, 48, 24), device="cuda") * 0.05
    out = torch.empty((n, 24), device="cuda")
    for _ in range(3):
        SP.sparse_conv(x, w, nbr, out=out)
    assert _lib.last_conv_kernel() == "spconv_direct16_kernel"
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _lib.load().eprecon_profile_mark_async(63, _lib.current_stream())      # marker: the 20 timed launches follow
    e0.record()
    for _ in range(20):
        SP.sparse_conv(x, w, nbr, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    flops = 2.0 * pairs * 48 * 24
    print(f"rows {n}  workgroups {(n + 127) // 128}  live pairs {pairs} ({live_frac:.3f} of 27 N)  {us:.1f} us per launch (HIP events, alone)")
    print(f"algorithmic {flops / 1e9:.2f} GFLOP -> {flops / (us * 1e-6) / 1e12:.1f} TFLOP/s = {flops / (us * 1e-6) / 1e12 / 157.3:.3f} of the fp32-MFMA peak; "
          f"output-stationary {2.0 * n * 27 * 48 * 24 / 1e9:.2f} GFLOP")


if __name__ == "__main__":
    with torch.no_grad():
        main()
