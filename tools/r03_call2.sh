#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c2
mkdir -p $O
cd $R
timeout 200 python -m pytest tests/test_dense_conv3d_gpu.py tests/test_training_gpu.py tests/test_occupancy_init_gpu.py -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -8 $O/tests.log
timeout 120 python tools/conv3d_probe.py 32 32 > $O/probe_32_32.txt 2>&1; cat $O/probe_32_32.txt
timeout 120 python tools/conv3d_probe.py 16 16 > $O/probe_16_16.txt 2>&1; cat $O/probe_16_16.txt
timeout 120 python tools/conv3d_probe.py 32 1 > $O/probe_32_1.txt 2>&1; cat $O/probe_32_1.txt
