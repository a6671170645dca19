#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02h
mkdir -p $O
cd $R
for v in 1 0 1 0; do
  EPRECON_CONV_WIDE=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 wide=$v', round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms']*1e3,1), round(d['roofline_conv']['avg_launch_ms']*1e3,1))"
done
for v in 1 0 1 0; do
  EPRECON_CONV_WIDE=$v timeout 300 python bench.py --workload cfg4 --steps 24 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 wide=$v', round(d['ms_per_step'],3))"
done
