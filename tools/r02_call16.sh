#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_sparse_gpu.py tests/test_dense2d_gpu.py -m gpu -x -q 2>&1 | tail -3
REPS=50 timeout 200 python tools/conv_only.py 2>&1 | tail -1
timeout 200 python tools/conv_shapes_ab.py pfB 2>&1 | tail -20
timeout 300 python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', round(d['ms_per_step'],3), 'conv us', round(d['roofline_conv']['avg_launch_ms']*1e3,1), 'gather', round(d['roofline']['avg_launch_ms']*1e3,1))"
