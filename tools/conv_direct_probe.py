"""Where the time of the cfg4-leading convolution goes (3x3x3 48 -> 24 on the real kernel map of the scene's last fragment,
spconv_direct16_kernel): time against the number of rows (the first n rows of the map: startup, slope, tail) — and, run under the
ablation builds of csrc/sparse_conv_direct_impl.hpp (EP_DIRECT_ABL, EPRECON_LIB_PATH), the same launch without its gathers / weight
traffic / MFMAs.
    python tools/conv_direct_probe.py [--sweep]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eprecon_amd import sparse as SP  # noqa: E402
from eprecon_amd import torchsparse_utils as TU  # noqa: E402
from eprecon_amd.fragment_step import Cfg4Step  # noqa: E402
from conv_tail_ab import timed  # noqa: E402


def main():
    dev = torch.device("cuda")
    saved = "/tmp/eprecon_probe_nbr.pt"     # (ablation builds give wrong results: they time the map the library's own run left)
    if os.path.exists(saved):
        nbr = torch.load(saved).to(dev)
    else:
        step = Cfg4Step(seed=0, device=dev)
        for _ in range(2 * step.n_fragments):
            step.run()
        torch.cuda.synchronize()
        maps = [e.vset._k3 for e in TU._VOX_CACHE if e.vset._k3 is not None and e.vset._k3.shape[1] > 200000]
        nbr = max([m for m in maps if float((m >= 0).float().mean()) > 0.2], key=lambda m: m.shape[1])
        torch.save(nbr.cpu(), saved)
    n = nbr.shape[1]
    cin, cout = int(os.environ.get("PROBE_CIN", "48")), int(os.environ.get("PROBE_COUT", "24"))
    x = torch.randn((n, cin), device=dev)
    w = torch.randn((27, cin, cout), device=dev) * 0.05
    pairs = int((nbr >= 0).sum())
    tag = os.environ.get("EPRECON_LIB_PATH", "library")
    rows = [n]
    if "--sweep" in sys.argv:
        rows = [128 * 256 * r // 4 for r in (1, 2, 4, 6, 8, 10, 12, 16, 20, 24, 28, 32, 36)] + [n]
    for m in rows:
        sub = nbr[:, :m].contiguous()
        out = torch.empty((m, cout), device=dev)
        run = lambda: SP.sparse_conv(x, w, sub, out=out)
        for _ in range(3):
            run()
        us = min(timed(run), timed(run))
        live = int((sub >= 0).sum())
        print(f"{tag}: rows {m:7d} workgroups {(m + 127) // 128:5d} ({(m + 127) // 128 / 256:5.2f} per CU)  {us:7.1f} us  "
              f"{2.0 * live * cin * cout / us / 1e6:6.1f} TF live  {us / ((m + 127) // 128) * 256:6.2f} us per workgroup-per-CU")


if __name__ == "__main__":
    with torch.no_grad():
        main()
