"""Interleaved A/B of the round-5 switches on cfg4 in ONE process: the variants take turns (one pass of the scene each, many
rounds), so that the drift between separate runs — 13.5 to 14.1 ms per fragment for the SAME build on one box (gpurun r05d) —
cancels.  Prints mean +- standard error of the per-pass times (ms per fragment) and the paired difference to the default.
    python tools/ab_cfg4.py [rounds]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eprecon_amd.modules as M  # noqa: E402
import eprecon_amd.neucon_network as NN  # noqa: E402
from eprecon_amd import _lib  # noqa: E402
from eprecon_amd.fragment_step import Cfg4Step  # noqa: E402

VARIANTS = [("default", {}),
            ("EPRECON_SPVCNN_NATIVE=0", {"native": False}),
            ("EPRECON_PREFETCH=0", {"prefetch": False}),
            ("both off (the round-4 issue order)", {"native": False, "prefetch": False})]


def apply(cfg):
    M._NATIVE_SPVCNN = cfg.get("native", True)
    NN._PREFETCH = cfg.get("prefetch", True)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    step = Cfg4Step(seed=0, device=torch.device("cuda"), pipeline=False)
    for _ in range(2 * step.n_fragments):
        step.run()
    times = {name: [] for name, _ in VARIANTS}
    reads = {}
    for r in range(rounds):
        order = VARIANTS if r % 2 == 0 else VARIANTS[::-1]      # (alternate the order: no variant always runs behind the same one)
        for name, cfg in order:
            apply(cfg)
            step.k = 0
            torch.cuda.synchronize()
            r0 = _lib.HOST_READS
            t0 = time.perf_counter()
            for _ in range(step.n_fragments):
                step.run()
            torch.cuda.synchronize()
            times[name].append((time.perf_counter() - t0) / step.n_fragments * 1e3)
            reads[name] = (_lib.HOST_READS - r0) / step.n_fragments
    apply({})
    base = np.array(times["default"])
    print(f"# cfg4 unpipelined, {rounds} interleaved passes of the {step.n_fragments}-fragment scene per variant (ms per fragment)")
    for name, _ in VARIANTS:
        t = np.array(times[name])
        d = t - base
        print(f"{name:40s} {t.mean():7.3f} +- {t.std(ddof=1) / np.sqrt(len(t)):.3f}   reads {reads[name]:4.1f}   "
              f"vs default {d.mean():+.3f} +- {d.std(ddof=1) / np.sqrt(len(d)):.3f}")


if __name__ == "__main__":
    with torch.no_grad():
        main()
