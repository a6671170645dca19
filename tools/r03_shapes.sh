#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_shapes
rm -rf $O; mkdir -p $O
cd $R
run() { tag=$1; shift; env "$@" timeout 300 python tools/conv_shapes_ab.py $tag > $O/$tag.txt 2>&1; tail -1 $O/$tag.txt; }
run lean A=1
run g1 EPRECON_CONV_DIRECT_G1=1
paste -d'|' <(cut -c1-62 $O/lean.txt) <(cut -c52-62 $O/g1.txt) > $O/table.txt
grep -v "SPVCNN0\|s0" $O/table.txt
