#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02f
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_sparse_gpu.py tests/test_spvcnn_gpu.py tests/test_dense2d_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 200 python tools/conv_shapes_ab.py new > $O/ab_new.txt 2>&1
EPRECON_CONV_WIDE=0 timeout 200 python tools/conv_shapes_ab.py old > $O/ab_old.txt 2>&1
EPRECON_CONV_SPLITK=0 timeout 200 python tools/conv_shapes_ab.py new_nosplitk > $O/ab_new_nosplitk.txt 2>&1
paste -d'|' <(cut -c1-75 $O/ab_new.txt) <(cut -c48-75 $O/ab_old.txt) <(cut -c48-75 $O/ab_new_nosplitk.txt)
