#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_sparse_gpu.py tests/test_spvcnn_gpu.py tests/test_dense2d_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 200 python tools/conv_shapes_ab.py pipe 2>&1 | grep -v amdgpu > /tmp/a.txt
EPRECON_CONV_SPLITK_PIPE=0 timeout 200 python tools/conv_shapes_ab.py nopipe 2>&1 | grep -v amdgpu > /tmp/b.txt
paste -d'|' <(cut -c1-75 /tmp/a.txt) <(cut -c48-75 /tmp/b.txt)
for v in 1 0 1 0; do
  EPRECON_CONV_SPLITK_PIPE=$v timeout 300 python bench.py --workload cfg4 --steps 24 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 splitk_pipe=$v', round(d['ms_per_step'],3))"
done
for v in 1 0; do
EPRECON_CONV_SPLITK_PIPE=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 splitk_pipe=$v', round(d['ms_per_step'],3))"
done
