"""gpurun_out/<dir> (tools/collect_profiles_r04.sh) -> the committed summaries under profiles/<round>/"""
import collections, csv, json, os, sys

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)
bench = [l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1]
open(os.path.join(dst, f"bench_cfg2_{tag}.json"), "w").write(bench)
b = json.loads(bench)
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 35
rows = list(csv.DictReader(open(os.path.join(src, "stats", "r_kernel_stats.csv"))))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = [f"# MI355X, {tag}: rocprofv3 --kernel-trace --stats -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline",
       f"# bench line of the same build (no profiler): {b['ms_per_step']:.3f} ms/step, {b['value']:.1f} fragments/s, "
       f"dominant kernel {b['roofline']['avg_launch_ms'] * 1e3:.1f} us (HIP events) -> {b['roofline']['achieved']:.0f} GB/s",
       "# kernel | calls | calls/step | total_ms | avg_us | pct"]
for r in rows:
    out.append(f"{r['Name'][:110]} | {r['Calls']} | {int(r['Calls']) / steps:.1f} | {float(r['TotalDurationNs']) / 1e6:.2f} | "
               f"{float(r['AverageNs']) / 1e3:.2f} | {float(r['Percentage']):.2f}")
out.append(f"# total kernel time {tot / 1e6:.1f} ms over {steps} steps = {tot / 1e6 / steps:.3f} ms/step "
           f"(sum over concurrent streams; wall time per step is lower)")
out.append("")
out.append("# PMC passes (separate runs, --kernel-trace + --pmc only; bench.py --steps 3 --warmup 1): average per dispatch")
pmc = {}
for d in sorted(os.listdir(src)):
    f = os.path.join(src, d, "r_counter_collection.csv")
    if not d.startswith("pmc_") or not os.path.exists(f):
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "bp_gather_mlp_kernel<256" in n or "spconv_resident_kernel<1, true, 4" in n or "conv3d_tile16_kernel<2, 2>" in n:
            short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            acc[(short, r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (n, g, c), v in sorted(acc.items()):
        out.append(f"{n} | grid {g} | {c} | {sum(v) / len(v):.1f} | n={len(v)}")
        if "bp_gather" in n:
            pmc[c] = sum(v) / len(v)
open(os.path.join(dst, f"bench_cfg2_{tag}_rocprof_summary.txt"), "w").write("\n".join(out) + "\n")
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    # gfx950: FETCH_SIZE tallies wide coalesced reads at 1/2 (MI355X_MICROARCH.md, HBM section) -> x2; KiB units
    traffic = 2 * pmc["FETCH_SIZE"] * 1024 + pmc["WRITE_SIZE"] * 1024
    import re
    rnd = re.search(r"r\d+", os.path.basename(os.path.normpath(dst))).group(0)   # the scratch copy is profiles_rNN, committed as profiles/rNN
    rec = {"kernel": "bp_gather_mlp_kernel<256,MEAN,6,1>", "fetch_size_kib_raw": pmc["FETCH_SIZE"],
           "write_size_kib": pmc["WRITE_SIZE"], "fetch_correction": "x2 (gfx950 wide-read tally)",
           "traffic_bytes": traffic, "l1_accesses": pmc.get("TCP_TOTAL_CACHE_ACCESSES_sum"),
           "l1_to_l2_read_requests": pmc.get("TCP_TCC_READ_REQ_sum"), "l2_hit": pmc.get("TCC_HIT_sum"),
           "l2_miss": pmc.get("TCC_MISS_sum"), "source": f"profiles/{rnd}/bench_cfg2_{tag}_rocprof_summary.txt"}
    json.dump(rec, open(os.path.join(dst, "pmc_traffic_bp_gather.json"), "w"), indent=1)
    print(rec)
print("\n".join(out[:14]))
