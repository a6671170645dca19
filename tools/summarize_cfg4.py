"""rocprofv3 --kernel-trace --stats output of `bench.py --workload cfg4` -> committed summaries:
    profiles/<round>/bench_cfg4_rocprof_summary.txt   per-kernel table (per fragment)
    profiles/<round>/cfg4_kernel_stats.json           launches / kernel time per steady-state fragment
Usage: python tools/summarize_cfg4.py <dir with r_kernel_stats.csv + r_kernel_trace.csv> <profiles/rNN> <bench json line file>
Fragments are delimited in the trace by `init_mark_kernel` (one launch per NeuConNet.forward); the first
fragments (calibration, warm-up, first-use allocations) are skipped."""
import csv
import json
import os
import sys

src, dst, bench_file = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)
bench = json.loads([l for l in open(bench_file) if l.startswith("{")][-1])
trace = list(csv.DictReader(open(os.path.join(src, "r_kernel_trace.csv"))))
trace.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(trace) if "init_mark" in r["Kernel_Name"]]
steady = list(zip(marks[-9:-1], marks[-8:]))      # the last 8 complete fragments
per_frag, busy = [], []
names = {}
for a, b in steady:
    seg = trace[a:b]
    per_frag.append(len(seg))
    busy.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e6)
    for r in seg:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:100]
        c = names.setdefault(n, [0, 0])
        c[0] += 1
        c[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
nf = len(steady)
rec = {"launches_per_fragment": round(sum(per_frag) / nf, 1), "kernel_ms_per_fragment": round(sum(busy) / nf, 3),
       "fragments_sampled": nf, "bench_ms_per_fragment_same_build": bench["ms_per_step"],
       "command": "rocprofv3 --kernel-trace --stats -- python bench.py --workload cfg4 --steps 12 --warmup 4"}
json.dump(rec, open(os.path.join(dst, "cfg4_kernel_stats.json"), "w"), indent=1)
out = [f"# MI355X: {rec['command']}",
       f"# bench line of the same build without the profiler: {bench['ms_per_step']:.2f} ms/fragment",
       f"# steady state ({nf} fragments): {rec['launches_per_fragment']} launches and {rec['kernel_ms_per_fragment']} ms of "
       "kernel time per fragment",
       "# kernel | launches/fragment | ms/fragment | avg_us"]
for n, (c, t) in sorted(names.items(), key=lambda kv: -kv[1][1]):
    out.append(f"{n} | {c / nf:.1f} | {t / nf / 1e6:.3f} | {t / c / 1e3:.1f}")
open(os.path.join(dst, "bench_cfg4_rocprof_summary.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:40]))
