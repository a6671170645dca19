"""Join of tools/trace_cfg4_layers.py's outputs: <dir>/r_kernel_trace.csv (rocprofv3), <dir>/conv.log (EPRECON_CONV_LOG),
<dir>/stages.json -> per-stage launches / kernel time and one line per convolution launch of the last fragment.
    python tools/summarize_cfg4_layers.py <dir> > profiles/rNN/cfg4_layers.txt"""
import collections
import csv
import glob
import json
import os
import re
import sys

src = sys.argv[1]
stages = json.load(open(os.path.join(src, "stages.json")))
trace = list(csv.DictReader(open(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0])))
trace.sort(key=lambda r: int(r["Start_Timestamp"]))
log = [l.split() for l in open(os.path.join(src, "conv.log"))]
# one log line per convolution CALL; the cross-workgroup split-K kernel is two launches (spconv_wide_kernel + its
# spconv_wide_reduce_kernel): the reduce launch is folded into the call's time, not counted as a call of its own
is_conv = lambda n: re.search(r"spconv_|conv3d_tile|conv2d_tile", n) is not None
is_tail = lambda n: "spconv_wide_reduce_kernel" in n
order = sorted(trace, key=lambda r: int(r["Dispatch_Id"])) if "Dispatch_Id" in trace[0] else trace   # host launch order = the log's
convs = []
for r in order:
    if not is_conv(r["Kernel_Name"]):
        continue
    if is_tail(r["Kernel_Name"]):
        assert convs and "spconv_wide_kernel" in convs[-1]["Kernel_Name"], "a reduce launch without its split-K launch"
        convs[-1]["tail_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        continue
    convs.append(r)
assert len(convs) == len(log), (len(convs), len(log))
for r, l in zip(convs, log):
    r["layer"] = l
bounds = [i for i, r in enumerate(trace) if "profile_mark" in r["Kernel_Name"] and int(r["Grid_Size_X"]) == 64]
frags = list(zip(bounds[-5:-1], bounds[-4:]))        # the last four fragments (one pass over the scene)
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
short = lambda n: re.sub(r"\(anonymous namespace\)::|void ", "", n).split("(")[0][:70]
per_stage = collections.OrderedDict()
kernels = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for a, b in frags:
    stack = []
    for r in trace[a + 1:b]:
        if "profile_mark" in r["Kernel_Name"]:
            sid = int(r["Grid_Size_X"]) // 64 - 1 - 10
            if sid % 2 == 0:
                stack.append(stages[sid // 2])
            else:
                stack.pop()
            continue
        name = stack[-1] if stack else "(between stages)"
        if name.startswith("convgru"):
            pass
        acc = per_stage.setdefault(name, [0, 0.0])
        acc[0] += 1
        acc[1] += dur(r)
        k = kernels[name][short(r["Kernel_Name"])]
        k[0] += 1
        k[1] += dur(r)
nf = len(frags)
print(f"# per fragment, mean of the last {nf} fragments (unpipelined, markers add {2 * len(stages)} empty launches that are not counted)")
print(f"# total: {sum(v[0] for v in per_stage.values()) / nf:.0f} launches, {sum(v[1] for v in per_stage.values()) / nf / 1e3:.2f} ms of kernel time")
print("# stage | launches | kernel ms   (gru_fusion = the bookkeeping around the ConvGRUs)")
for k, (n, t) in per_stage.items():
    print(f"{k:18s} {n / nf:7.1f} {t / nf / 1e3:8.3f}")
print("\n# top kernels per stage (launches, ms per fragment)")
for k in per_stage:
    print(f"[{k}]")
    for kn, (n, t) in sorted(kernels[k].items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"    {n / nf:6.1f} {t / nf / 1e3:7.3f}  {kn}")
print("\n# convolution launches of the last fragment: stage | rows K Cin Cout | kernel | us")
a, b = frags[-1]
stack = []
for r in trace[a + 1:b]:
    if "profile_mark" in r["Kernel_Name"]:
        sid = int(r["Grid_Size_X"]) // 64 - 1 - 10
        stack.append(stages[sid // 2]) if sid % 2 == 0 else stack.pop()
        continue
    if "layer" in r:
        l = r["layer"]
        n, K, ci, co = map(int, l[:4])
        gf = 2.0 * n * K * ci * co / 1e9
        us = dur(r) + r.get("tail_ns", 0) / 1e3
        print(f"{(stack[-1] if stack else '-'):16s} {n:7d} {K:2d} {ci:4d} {co:4d}  {l[4]:30s} {' '.join(l[5:]):22s} {us:7.1f} us  "
              f"({gf / us * 1e3:5.1f} TF dense-equivalent)")
