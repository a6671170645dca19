#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02j
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_back_project_gpu.py tests/test_occupancy_init_gpu.py tests/test_neucon_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
for v in 1 0 1 0; do
  EPRECON_BP_DENSE=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 dense=$v', round(d['ms_per_step'],3), 'gather us', round(d['roofline']['avg_launch_ms']*1e3,1), d['roofline']['kernel'][:40])"
done
