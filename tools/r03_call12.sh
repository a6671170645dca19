#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c12
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_dense_conv3d_gpu.py tests/test_bn_fused_gpu.py -x -q > $O/new.log 2>&1; echo "new rc=$?" >> $O/new.log
tail -12 $O/new.log | cut -c1-220
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "LV=2 32 32 dense" "LV=2 32 32 dense_bias" "LV=2 32 16 dense" "LV=2 16 16 dense" "LV=2 16 16 dense_bias" "LV=0 32 32 gather" "LV=0 32 16 gather" "LV=0 16 16 gather"; do
  set -- $cfg
  lv=${1#LV=}; cin=$2; cout=$3; mode=$4
  i=$((i+1)); d=$O/p$i
  EPRECON_CONV_DENSE3D=$lv timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python $R/tools/conv3d_probe.py $cin $cout 20 $mode > $d.log 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  line=$(python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if ("conv3d_tile" in r["Name"] or "spconv_" in r["Name"]) and int(r["Calls"]) >= 20]
rows.sort(key=lambda r: -int(r["Calls"]))
for r in rows[:2]:
    n = r['Name']
    k = n[n.find('conv3d'):][:40] if 'conv3d' in n else n[n.find('spconv'):][:44]
    print(f"{k} calls {r['Calls']} avg {float(r['AverageNs'])/1e3:.1f} us min {float(r['MinNs'])/1e3:.1f} us;", end=" ")
PY
)
  echo "$cfg | $line" | tee -a $O/durations.txt
  find $d -name "*kernel_trace.csv" -delete
done
cd $R
for lv in 1 2; do
EPRECON_CONV_DENSE3D=$lv python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('DENSE3D=$lv ms_per_step', b['ms_per_step'], b['roofline_conv']['kernel'][:30], b['roofline_conv']['avg_launch_ms'], b['roofline_conv']['frac'])" | tee -a $O/durations.txt
done
