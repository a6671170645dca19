"""Where the HOST time of a cfg4 fragment goes (cProfile over steady-state fragments, unpipelined): the functions with the
largest own time, and the blocking reads.  The GPU keeps running while the host queues, so a fragment is as slow as the
slower of the two sides in each stretch between two reads.
    python tools/profile_cfg4_host.py [fragments]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eprecon_amd.fragment_step import Cfg4Step  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    step = Cfg4Step(seed=0, device=torch.device("cuda"), pipeline=False)
    for _ in range(2 * step.n_fragments):
        step.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step.run()
    torch.cuda.synchronize()
    print(f"free-running {(time.perf_counter() - t0) / n * 1e3:.2f} ms/fragment")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        step.run()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime")
    rows = []
    for (fn, line, name), (cc, nc, tt, ct, _) in st.stats.items():
        rows.append((tt / n * 1e3, ct / n * 1e3, nc / n, f"{os.path.basename(fn)}:{line}:{name}"))
    rows.sort(reverse=True)
    print(f"# own ms/fragment | cumulative ms/fragment | calls/fragment | function   (profiler overhead included: {sum(r[0] for r in rows):.1f} ms total)")
    for r in rows[:60]:
        print(f"{r[0]:8.3f} {r[1]:8.3f} {r[2]:8.1f}  {r[3]}")


if __name__ == "__main__":
    with torch.no_grad():
        main()
