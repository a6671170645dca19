#!/bin/bash
# per-stage / per-layer kernel accounting of a cfg4 fragment (stage markers + convolution log joined with the rocprofv3
# kernel trace) -> gpurun_out/$1/cfg4_layers.txt.  EPRECON_NO_GRAPH=1: launches replayed from a HIP graph (the 2D fusion stack, the
# decoder's query side) are not seen by the convolution log, so the accounting run issues them one by one
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r04_layers}
L=$O/layers; rm -rf $L; mkdir -p $L
cd /tmp && export TMPDIR=/tmp
EPRECON_NO_GRAPH=1 EPRECON_CONV_LOG=$L/conv.log rocprofv3 --kernel-trace --output-format csv -d $L -o r -- python $R/tools/trace_cfg4_layers.py $L > $L/run.log 2>&1
cd $R
python tools/summarize_cfg4_layers.py $L > $O/cfg4_layers.txt 2> $L/sum.err
tail -3 $L/sum.err
rm -f $L/r_kernel_trace.csv $L/*/*kernel_trace.csv
head -40 $O/cfg4_layers.txt
