#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r04_dec}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/tools/profile_decoder.py 30 > $O/decoder_times.txt 2>&1
cat $O/decoder_times.txt | tail -3
EPRECON_DECODER_ONLY_FUSED=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python $R/tools/profile_decoder.py 10 > $O/stats.log 2>&1
rm -f $O/stats/r_kernel_trace.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/stats/r_kernel_stats.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6, "launches", sum(int(r["Calls"]) for r in rows))
for r in rows[:45]:
    print(f'{r["Name"][:90]:90s} {r["Calls"]:>6s} {float(r["TotalDurationNs"])/1e6:8.3f} ms {float(r["AverageNs"])/1e3:8.1f} us')
PY
