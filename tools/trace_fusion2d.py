"""Timeline of the 2D fusion stack of the occupancy initialisation (feat_fusion_pre, HIP-graph replay) under
rocprofv3 --kernel-trace: which queue (stream) carries the critical path and what runs on it.
    rocprofv3 --kernel-trace --output-format csv -d D -o r -- python tools/trace_fusion2d.py
    python tools/trace_fusion2d.py --summarize D/.../r_kernel_trace.csv"""
import csv
import os
import sys


def summarize(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "profile_mark" in r["Kernel_Name"] and int(r["Grid_Size_X"]) in (62 * 64, 63 * 64)]
    a, b = marks[-2], marks[-1]
    seg = rows[a + 1:b]
    t0 = int(rows[a]["End_Timestamp"])
    qkey = "Queue_Id" if "Queue_Id" in seg[0] else "Stream_Id"
    print(f"# one replay of feat_fusion_pre: {len(seg)} kernels, wall {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, "
          f"sum of durations {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e3:.1f} us")
    print("# start us | dur us | gap before (same queue) us | queue | kernel | grid")
    last_end = {}
    for r in seg:
        s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[qkey]
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("epconv::", "").replace("void ", "")[:48]
        print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {gap:7.1f}  q{q:>3s}  {name:48s} {r['Grid_Size_X']}")


def main():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from eprecon_amd import _lib
    from eprecon_amd.fragment_step import Cfg2Step
    step = Cfg2Step(seed=0)
    net = step.init_net
    f = step.features_init
    for _ in range(5):
        step.run()
    torch.cuda.synchronize()
    lib = _lib.load()
    args = [torch.stack([v[lvl][0] for v in f]) for lvl in (2, 1, 0)]
    for _ in range(3):
        net.feat_fusion_pre(*args)
    torch.cuda.synchronize()
    lib.eprecon_profile_mark_async(61, _lib.current_stream())
    net.feat_fusion_pre(*args)
    lib.eprecon_profile_mark_async(62, _lib.current_stream())
    torch.cuda.synchronize()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
        summarize(sys.argv[2])
    else:
        import torch
        with torch.no_grad():
            main()
