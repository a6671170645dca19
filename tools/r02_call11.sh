#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02k
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gru_fusion_gpu.py tests/test_cfg4_gpu.py tests/test_sparse_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
for v in 1 0 1 0; do
  EPRECON_GRU_STREAMS=$v timeout 300 python bench.py --workload cfg4 --steps 24 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 gru_streams=$v', round(d['ms_per_step'],3))"
done
