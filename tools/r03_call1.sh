#!/bin/bash
# round 3, call 1: new dense-grid conv tests, full GPU suite, bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c1
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_dense_conv3d_gpu.py -x -q > $O/dense.log 2>&1; echo "dense rc=$?" >> $O/dense.log
tail -15 $O/dense.log
timeout 600 python -m pytest tests -m gpu -q -x --deselect tests/test_dense_conv3d_gpu.py > $O/gpu.log 2>&1; echo "gpu rc=$?" >> $O/gpu.log
tail -15 $O/gpu.log
timeout 400 python bench.py --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cat $O/bench.json | head -c 6000
tail -5 $O/bench.err
