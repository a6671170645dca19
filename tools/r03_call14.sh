#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c14
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_dense_conv3d_gpu.py tests/test_bn_fused_gpu.py tests/test_occupancy_init_gpu.py tests/test_pins_gpu.py -x -q > $O/new.log 2>&1; echo "new rc=$?" >> $O/new.log
tail -5 $O/new.log | cut -c1-220
cd /tmp && export TMPDIR=/tmp
for cfg in "ABL=0 32 32 dense" "ABL=0 32 32 dense_bias" "ABL=1 32 32 dense_bias" "ABL=0 32 16 dense" "ABL=0 16 16 dense" "ABL=0 48 32 dense_bias" "ABL=0 64 32 dense_bias"; do
  set -- $cfg
  abl=${1#ABL=}; cin=$2; cout=$3; mode=$4
  d=$O/p
  EPRECON_D3_ABLATE=$abl timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python $R/tools/conv3d_probe.py $cin $cout 20 $mode > $d.log 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  python - "$f" "$cfg" <<'PY' | tee -a $O/durations.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "conv3d_tile16" in r["Name"] and int(r["Calls"]) >= 20]
for r in rows[:1]:
    print(sys.argv[2], "|", r['Name'][r['Name'].find('conv3d'):][:30], f"avg {float(r['AverageNs'])/1e3:.1f} us min {float(r['MinNs'])/1e3:.1f} us")
PY
  rm -rf $d
done
cd $R
python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('cfg2 ms_per_step', b['ms_per_step'], b['roofline_conv']['kernel'][:30], b['roofline_conv']['avg_launch_ms'], b['roofline_conv']['frac'])" | tee -a $O/durations.txt
python tools/profile_cfg2_stages.py 2>/dev/null | tail -16 | tee -a $O/durations.txt
