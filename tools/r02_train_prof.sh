#!/bin/bash
# training-step timing + kernel stats (round 2)
export TMPDIR=/tmp
mkdir -p gpurun_out/train
PYTHONPATH=. python tools/profile_train_step.py 10 2>&1 | grep -v "NCCL\|RCCL" | tail -3 > gpurun_out/train/wall.txt
cat gpurun_out/train/wall.txt
PYTHONPATH=. timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/train/prof -o train -- python tools/profile_train_step.py 5 > gpurun_out/train/prof.log 2>&1
f=$(find gpurun_out/train/prof -name "*kernel_stats.csv" | head -1)
[ -z "$f" ] && { tail -20 gpurun_out/train/prof.log; find gpurun_out/train/prof | head; exit 1; }
find gpurun_out/train/prof -name "*kernel_trace.csv" -delete
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}%  {int(r['Calls']):6d} x {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:110]}")
PY
