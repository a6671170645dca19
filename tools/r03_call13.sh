#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c13
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/gpu.log 2>&1; echo "gpu rc=$?" >> $O/gpu.log
tail -8 $O/gpu.log | cut -c1-250
cd /tmp && export TMPDIR=/tmp
for cfg in "ABL=1 32 32 dense" "ABL=1 32 32 dense_bias" "ABL=1 16 16 dense_bias" "ABL=4 32 32 dense_bias"; do
  set -- $cfg
  abl=${1#ABL=}; cin=$2; cout=$3; mode=$4
  d=$O/p_$abl_$cin_$cout_$mode
  EPRECON_D3_ABLATE=$abl timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python $R/tools/conv3d_probe.py $cin $cout 20 $mode > $d.log 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  python - "$f" "$cfg" <<'PY' | tee -a $O/ablate.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "conv3d_tile16" in r["Name"] and int(r["Calls"]) >= 20]
for r in rows[:1]:
    print(sys.argv[2], "|", r['Name'][r['Name'].find('conv3d'):][:30], f"avg {float(r['AverageNs'])/1e3:.1f} us min {float(r['MinNs'])/1e3:.1f} us")
PY
  rm -rf $d
done
