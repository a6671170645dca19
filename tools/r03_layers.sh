#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_layers
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
EPRECON_NO_GRAPH=1 EPRECON_CONV_LOG=$O/conv.log timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O -o r -- python tools/trace_cfg4_layers.py $O > $O/run.log 2>&1
echo "rc=$?" >> $O/run.log
grep -v "simple_timer\|rocpd" $O/run.log | tail -4 | cut -c1-300
python tools/summarize_cfg4_layers.py $O > $O/cfg4_layers.txt 2> $O/sum.err; tail -3 $O/sum.err
head -30 $O/cfg4_layers.txt
find $O -name "*kernel_trace.csv" -size +30M -delete
