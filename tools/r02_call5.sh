#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err; cat $O/bench_cfg2.json; tail -3 $O/bench_cfg2.err
timeout 300 python $R/bench.py --workload cfg4 --steps 24 --warmup 8 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -o r -- python $R/bench.py --workload cfg4 --steps 12 --warmup 4 > $O/stats_cfg4.log 2>&1
cd $R
python tools/summarize_cfg4.py $O/stats_cfg4 $O/profiles_r02 $O/bench_cfg4.json | head -50
rm -f $O/stats_cfg4/r_kernel_trace.csv
