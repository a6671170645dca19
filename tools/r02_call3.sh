#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_mask3dformer.py tests/test_neucon_gpu.py tests/test_gru_fusion_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python tools/profile_cfg4_stages.py 3 > $O/stages_cfg4.txt 2>&1; cat $O/stages_cfg4.txt
timeout 300 python tools/hostprof_cfg4.py 12 > $O/hostprof_cfg4.txt 2>&1; head -75 $O/hostprof_cfg4.txt
