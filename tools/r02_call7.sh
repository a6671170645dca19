#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_mask3dformer.py tests/test_spvcnn_gpu.py tests/test_gru_fusion_gpu.py tests/test_neucon_gpu.py tests/test_sparse_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 300 python bench.py --workload cfg4 --steps 24 --warmup 8 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; cat $O/bench_cfg4.json
timeout 300 python tools/profile_cfg4_stages.py 3 > $O/stages_cfg4.txt 2>&1; cat $O/stages_cfg4.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err; cat $O/bench_cfg2.json
