#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_sparse_gpu.py tests/test_spvcnn_gpu.py tests/test_gru_fusion_gpu.py tests/test_neucon_gpu.py tests/test_grid_ops_gpu.py tests/test_back_project_gpu.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do
  timeout 300 python bench.py --workload cfg4 --steps 24 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4', round(d['ms_per_step'],3))"
done
timeout 300 python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', round(d['ms_per_step'],3))"
