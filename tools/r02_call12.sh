#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02l
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_mask3dformer.py tests/test_neucon_gpu.py tests/test_cfg4_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
for v in 1 0 1 0; do
  EPRECON_DECODER_GRAPH=$v timeout 300 python bench.py --workload cfg4 --steps 24 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 decoder_graph=$v', round(d['ms_per_step'],3))"
done
timeout 300 python tools/profile_cfg4_stages.py 3 2>&1 | grep -v amdgpu | tail -24
