"""Times one 3x3x3 layer of the initialisation stack's shape on the dense-grid kernel and on the gather form, with the
dense kernel's phases switched off one at a time (EPRECON_D3_ABLATE, timing only):
    python tools/conv3d_probe.py [cin cout [reps]]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eprecon_amd import sparse as SP  # noqa: E402
from eprecon_amd.fragment_step import Cfg2Step  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    t = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps))
    return t[len(t) // 2], t[0]


def main():
    cin = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    cout = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    step = Cfg2Step(seed=0)
    out = step.run()
    coords = out["init"][1].contiguous()
    n = coords.shape[0]
    vs = SP.VoxelSet(coords, 2, dims=(48, 48, 48))
    dm, nbr = SP.DenseMap(vs, (48, 48, 48)), vs.kernel_map(3)
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((n, cin), device="cuda", generator=g)
    w = torch.randn((27, cin, cout), device="cuda", generator=g) / (27 * cin) ** 0.5
    b = torch.zeros(cout, device="cuda")
    lg, lb = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    pairs = int((nbr >= 0).sum())
    flops = 2.0 * pairs * cin * cout
    print(f"n={n} cin={cin} cout={cout} live pairs={pairs} flops={flops/1e9:.2f} G  (MFMA bound {flops/157.3e12*1e6:.1f} us)")
    y = torch.empty((n, cout), device="cuda")
    res = x if cin == cout else None

    def run(m):
        if cout == 1:
            return lambda: SP.sparse_conv_fused(x, w, m, b, out=y, bn_partial=True)
        return lambda: SP.sparse_conv_ln(x, w, m, b, lg, lb, 1e-5, out=y, relu=True, residual=res)
    if len(sys.argv) > 4:   # single variant, for rocprofv3 --kernel-trace --stats: dense | gather | dense_bias | gather_bias
        mode = sys.argv[4]
        m = dm if mode.startswith("dense") else nbr
        fn = (lambda: SP.sparse_conv(x, w, m, b, out=y)) if mode.endswith("bias") else run(m)
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return
    rows = [("gather form (kernel map)", nbr, None)]
    for name, abl in (("dense-grid kernel", "0"), ("  no MFMA loop", "1"), ("  no halo row loads", "4"), ("  neither", "5")):
        rows.append((name, dm, abl))
    for name, m, abl in rows:
        if abl is not None:
            os.environ["EPRECON_D3_ABLATE"] = abl
        med, mn = timed(run(m), reps)
        print(f"{name:28s} median {med:7.1f} us  min {mn:7.1f} us  -> {flops/med/1e6:6.1f} TF")
    os.environ["EPRECON_D3_ABLATE"] = "0"
    # plain variant (bias only, no LayerNorm) and BatchNorm-summary variant
    if cout > 1:
        for name, fn in (("dense, bias only", lambda: SP.sparse_conv(x, w, dm, b, out=y)),
                         ("dense, BN summaries", lambda: SP.conv_stats(x, w, dm, out=y)),
                         ("gather, bias only", lambda: SP.sparse_conv(x, w, nbr, b, out=y))):
            med, mn = timed(fn, reps)
            print(f"{name:28s} median {med:7.1f} us  min {mn:7.1f} us")


if __name__ == "__main__":
    main()
