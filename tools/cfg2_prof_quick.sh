R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06i
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $O/stats.log 2>&1
rm -f $O/stats/r_kernel_trace.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/stats/r_kernel_stats.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms/step", tot/1e6/35, "launches/step", sum(int(r["Calls"]) for r in rows)/35)
for r in rows[:60]:
    print(f'{r["Name"][:100]:100s} {int(r["Calls"])/35:6.1f} {float(r["TotalDurationNs"])/1e3/35:8.1f} us/step {float(r["AverageNs"])/1e3:8.1f} us')
PY
