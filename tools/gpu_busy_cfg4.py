"""How much of a cfg4 fragment's wall time the GPU is busy (unpipelined): run under rocprofv3 --kernel-trace, the steady-state loop
is bracketed by two marker launches; `--summarize <trace.csv>` then prints, per fragment, the wall time between the markers, the
union of all kernel intervals (all streams) and the sum of kernel durations.
    rocprofv3 --kernel-trace --output-format csv -d D -o r -- python tools/gpu_busy_cfg4.py
    python tools/gpu_busy_cfg4.py --summarize D/.../r_kernel_trace.csv"""
import csv
import os
import sys

N = 8


def summarize(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "profile_mark" in r["Kernel_Name"] and int(r["Grid_Size_X"]) in (62 * 64, 63 * 64)]
    a, b = marks[-2], marks[-1]
    seg = rows[a + 1:b]
    t0, t1 = int(rows[a]["End_Timestamp"]), int(rows[b]["Start_Timestamp"])
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
    busy, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += (cur_e - cur_s) if cur_e is not None else 0
    total = sum(e - s for s, e in iv)
    gaps = sorted(((iv[i + 1][0] - max(x[1] for x in iv[:i + 1][-8:])) for i in range(len(iv) - 1)), reverse=True)
    print(f"# cfg4 unpipelined under rocprofv3 --kernel-trace, {N} steady-state fragments between two markers")
    print(f"wall {(t1 - t0) / N / 1e6:.2f} ms/fragment   GPU busy (union over streams) {busy / N / 1e6:.2f} ms   "
          f"sum of kernel durations {total / N / 1e6:.2f} ms   launches {len(seg) / N:.0f}   idle {(t1 - t0 - busy) / N / 1e6:.2f} ms/fragment")
    # CU-level occupancy: a launch of W workgroups covers at most min(W, 256) of the 256 CUs while it runs;
    # sum over launches of that share x duration, against 256 CUs x wall — what "busy" leaves out: ~600 of a fragment's launches
    # are 3-30 us kernels on <= 10k rows that occupy a fraction of the chip
    def wgs(r):
        g = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        w = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
        return max(1, g // max(w, 1))
    cu_time = sum(min(wgs(r), 256) / 256.0 * (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in seg)
    small = [r for r in seg if wgs(r) < 256]
    print(f"CU-level occupancy: sum over launches of min(workgroups, 256) / 256 x duration = {cu_time / N / 1e6:.2f} ms per fragment = "
          f"{cu_time / (t1 - t0):.2f} of 256 CUs x wall ({cu_time / max(busy, 1):.2f} of the busy time); launches with fewer than 256 "
          f"workgroups: {len(small) / N:.0f} per fragment, {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in small) / N / 1e6:.2f} ms")
    big = [g for g in gaps if g > 20000]
    print(f"idle gaps longer than 20 us: {len(big) / N:.1f} per fragment, {sum(big) / N / 1e6:.2f} ms; longer than 5 us: "
          f"{sum(1 for g in gaps if g > 5000) / N:.0f} per fragment, {sum(g for g in gaps if g > 5000) / N / 1e6:.2f} ms")


def main():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from eprecon_amd import _lib
    from eprecon_amd.fragment_step import Cfg4Step
    step = Cfg4Step(seed=0, device=torch.device("cuda"), pipeline=False)
    for _ in range(2 * step.n_fragments):
        step.run()
    torch.cuda.synchronize()
    lib = _lib.load()
    lib.eprecon_profile_mark_async(61, _lib.current_stream())
    for _ in range(N):
        step.run()
    lib.eprecon_profile_mark_async(62, _lib.current_stream())
    torch.cuda.synchronize()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
        summarize(sys.argv[2])
    else:
        import torch
        with torch.no_grad():
            main()
