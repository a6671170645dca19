#!/bin/bash
# quick A/B of a convolution change: parity tests, the per-shape list, the cfg4 instance, cfg4 / cfg2 bench
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r04_conv}; mkdir -p $O; cd $R
python -m pytest tests/test_sparse_gpu.py tests/test_dense_conv3d_gpu.py -x -q -m gpu 2>&1 | tail -3
python tools/conv_shapes_ab.py ${1:-r04_conv} 2>/dev/null > $O/conv_shapes.txt; tail -1 $O/conv_shapes.txt
python tools/conv_cfg4_instance.py 2>/dev/null | tail -2
EPRECON_CFG4_PIPELINE=0 python bench.py --workload cfg4 --steps 24 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4', d['ms_per_step'])"
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', d['ms_per_step'])"
