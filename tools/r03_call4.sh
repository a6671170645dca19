#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c4
mkdir -p $O
cd $R
timeout 200 python -m pytest tests/test_dense_conv3d_gpu.py tests/test_training_gpu.py -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -6 $O/tests.log
for wv in 2 4; do
  for sh in "32 32" "16 16" "32 16"; do
    echo "== WV=$wv $sh" | tee -a $O/probe.txt
    EPRECON_D3_WV=$wv timeout 120 python tools/conv3d_probe.py $sh 2>&1 | grep -v amdgpu.ids | tee -a $O/probe.txt
  done
done
