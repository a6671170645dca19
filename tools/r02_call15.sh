#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in 1 0 1 0; do
  EPRECON_SCAN_SMALL=$v timeout 300 python bench.py --workload cfg4 --steps 24 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 scan_small=$v', round(d['ms_per_step'],3))"
done
for v in 1 0; do
EPRECON_SCAN_SMALL=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 scan_small=$v', round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms']*1e3,1))"
done
