"""A/B of the 16 + 8 column split of the direct gather kernel (csrc/sparse_conv_direct_impl.hpp, TAIL form: the last 8 columns of
C_out = 16 m + 8 on v_mfma_f32_4x4x1_16B_f32) against the padded 16-column tile (EPRECON_CONV_TAIL8=0), in ONE process: the
shapes of tools/conv_shapes_ab.py with C_out = 8 / 24 / 40 on random 35 %-filled sets, and the cfg4-leading instance (48 -> 24 on
the real kernel map of the scene's last fragment).  Prints max |difference| of the two outputs (summation order differs) and
HIP-event times (20 launches, interleaved twice).
    python tools/conv_tail_ab.py [--instance]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eprecon_amd import _lib  # noqa: E402
from eprecon_amd import sparse as SP  # noqa: E402
from conv_shapes_ab import SHAPES, coords_for  # noqa: E402


def timed(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


MODES = [("padded", {"EPRECON_CONV_TAIL8": "0"}), ("16+8", {}), ("persist", {"EPRECON_CONV_PERSIST": "1"})]


def ab(what, x, w, nbr, aff=None, stats=True):
    """the same launch under the three forms (switches are read per launch): min of two interleaved timings each, the outputs
    (and the BatchNorm the summaries give) compared with the padded form"""
    n, co = x.shape[0], w.shape[2]
    outs, us, kern = {}, {m: [] for m, _ in MODES}, {}
    for rnd in range(2):
        for mode, env in MODES:
            for k in ("EPRECON_CONV_TAIL8", "EPRECON_CONV_PERSIST"):
                os.environ.pop(k, None)
            os.environ.update(env)
            out = torch.empty(n, co, device=x.device)
            run = (lambda: SP.conv_stats(x, w, nbr, out=out, in_affine=aff)) if stats else (lambda: SP.sparse_conv(x, w, nbr, out=out))
            for _ in range(3):
                res = run()
            us[mode].append(timed(run))
            kern[mode] = _lib.last_conv_kernel()
            if stats:
                y, partial = res
                sc, sh = SP.bn_affine(partial, torch.ones(co, device=x.device), torch.zeros(co, device=x.device), 1e-5)
                outs[mode] = (out.clone(), sc.clone(), sh.clone())
            else:
                outs[mode] = (out.clone(),)
    for k in ("EPRECON_CONV_TAIL8", "EPRECON_CONV_PERSIST"):
        os.environ.pop(k, None)
    ref = outs["padded"]
    scale = float(ref[0].abs().max())
    t = {m: min(v) for m, v in us.items()}
    line = f"{what:28s} N={n:7d} {w.shape[1]:4d}->{co:3d} "
    for m, _ in MODES:
        d = max(float((a - b).abs().max()) for a, b in zip(outs[m], ref))
        line += f" {m} {t[m]:7.1f} us ({t[m] / t['padded']:5.3f}, d {d:.1e}, {kern[m].replace('spconv_', '').replace('_kernel', '')})"
    print(line + f"  |out| {scale:.1f}")
    return [t[m] for m, _ in MODES]


def main():
    dev = torch.device("cuda")
    rng = np.random.default_rng(0)
    print("# " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith(("EPRECON_CONV", "EPRECON_LIB"))))
    tot = [0.0] * len(MODES)
    with torch.no_grad():
        for n, ci, co, what in SHAPES:
            if n < 40000 or co > 64:
                continue
            vs = SP.VoxelSet(torch.from_numpy(coords_for(n, rng)).to(dev), 1)
            nbr = vs.kernel_map(3)
            x = torch.randn(n, ci, device=dev)
            w = torch.randn(27, ci, co, device=dev) * 0.05
            tot = [a + b for a, b in zip(tot, ab(what, x, w, nbr))]
            if ci == 48 and co == 24:
                aff = (torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1, True)
                ab(what + " +BN/ReLU in", x, w, nbr, aff)
        print("sum " + "  ".join(f"{m} {v:.0f} us" for (m, _), v in zip(MODES, tot)))
        if "--instance" in sys.argv:
            from eprecon_amd import torchsparse_utils as TU
            from eprecon_amd.fragment_step import Cfg4Step
            step = Cfg4Step(seed=0, device=dev)
            for _ in range(2 * step.n_fragments):
                step.run()
            torch.cuda.synchronize()
            maps = [e.vset._k3 for e in TU._VOX_CACHE if e.vset._k3 is not None and e.vset._k3.shape[1] > 200000]
            cands = [m for m in maps if float((m >= 0).float().mean()) > 0.2]
            nbr = max(cands, key=lambda m: m.shape[1])
            n = nbr.shape[1]
            x = torch.randn((n, 48), device=dev)
            w = torch.randn((27, 48, 24), device=dev) * 0.05
            pairs = int((nbr >= 0).sum())
            t = ab("cfg4 instance (real map)", x, w, nbr, stats=False)
            fl = 2.0 * pairs * 48 * 24
            print("instance: live pairs %d; " % pairs + "  ".join(f"{m} {fl / v / 1e6 / 157.3:.3f}" for (m, _), v in zip(MODES, t)) + " of the fp32-MFMA peak")


if __name__ == "__main__":
    main()
