#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_2d
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
for v in 0 1; do
  EPRECON_CONV_DIRECT_2D=$v EPRECON_CONV_DIRECT_2D_MIN_ROWS=10000 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t$v -o r -- python tools/conv_layers_trace.py > $O/names$v.txt 2>$O/err$v.txt
  python tools/conv_layers_summary.py $O/t$v/r_kernel_trace.csv $O/names$v.txt > $O/sum$v.txt 2>&1
done
paste -d'|' <(cut -c1-120 $O/sum0.txt) <(cut -c24-120 $O/sum1.txt)
