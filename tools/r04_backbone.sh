#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r04_bb}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/tools/profile_backbone.py 20 2>/dev/null | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python $R/tools/profile_backbone.py 10 > $O/stats.log 2>&1
rm -f $O/stats/r_kernel_trace.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/stats/r_kernel_stats.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per call", tot/1e6/15, "launches per call", sum(int(r["Calls"]) for r in rows)/15)
for r in rows[:40]:
    print(f'{r["Name"][:100]:100s} {int(r["Calls"])/15:6.1f} {float(r["TotalDurationNs"])/1e6/15:8.3f} ms {float(r["AverageNs"])/1e3:8.1f} us')
PY
