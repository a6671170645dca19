#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in 1 0 1 0; do
  EPRECON_CFG2_BP_STREAM=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 bp_stream=$v', round(d['ms_per_step'],3), 'gather us', round(d['roofline']['avg_launch_ms']*1e3,1))"
done
