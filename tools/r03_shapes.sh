#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "EPRECON_CFG2_LEVELS_FIRST=1" "A=1" "EPRECON_CFG2_LEVELS_FIRST=1" "A=2" "EPRECON_CFG2_DEFER=0"; do
  env $v timeout 300 python bench.py --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cfg2 $v', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms']*1e3,1), round(d['roofline_conv']['avg_launch_ms']*1e3,1))"
done
timeout 600 python -m pytest tests/test_properties_gpu.py tests/test_occupancy_init_gpu.py tests/test_neucon_gpu.py -x -q 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
