#!/bin/bash
# true kernel durations (rocprofv3 --kernel-trace --stats) of the 3x3x3 layer variants
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # tag, env..., args
  tag=$1; shift
  env "$@" > /dev/null 2>&1
}
i=0
for cfg in "WV=4 ABL=0 32 32 dense" "WV=4 ABL=1 32 32 dense" "WV=4 ABL=5 32 32 dense" "WV=2 ABL=0 32 32 dense" "WV=4 ABL=0 32 32 dense_bias" "WV=4 ABL=1 32 32 dense_bias" "WV=4 ABL=0 32 32 gather" "WV=4 ABL=0 32 32 gather_bias" \
           "WV=4 ABL=0 16 16 dense" "WV=4 ABL=1 16 16 dense" "WV=2 ABL=0 16 16 dense" "WV=4 ABL=0 16 16 gather" "WV=4 ABL=0 32 16 dense" "WV=4 ABL=0 32 16 gather" "WV=4 ABL=0 32 1 dense" "WV=4 ABL=0 32 1 gather"; do
  set -- $cfg
  wv=${1#WV=}; abl=${2#ABL=}; cin=$3; cout=$4; mode=$5
  i=$((i+1))
  d=$O/p$i
  EPRECON_D3_WV=$wv EPRECON_D3_ABLATE=$abl timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python $R/tools/conv3d_probe.py $cin $cout 20 $mode > $d.log 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  line=$(python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if ("conv3d_tile" in r["Name"] or "spconv_" in r["Name"]) and int(r["Calls"]) >= 20]
rows.sort(key=lambda r: -int(r["Calls"]))
r = rows[0]
print(f"{r['Name'].split('(')[0][-60:]} calls {r['Calls']} avg {float(r['AverageNs'])/1e3:.1f} us min {float(r['MinNs'])/1e3:.1f} us")
PY
)
  echo "$cfg | $line" | tee -a $O/durations.txt
  find $d -name "*kernel_trace.csv" -delete
done
