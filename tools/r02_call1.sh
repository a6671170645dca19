#!/bin/bash
# round 2, GPU call 1: new parity tests + fresh cfg2 / cfg4 baselines and kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 300 python $R/bench.py --workload cfg4 --steps 24 --warmup 8 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -o r -- python $R/bench.py --workload cfg4 --steps 12 --warmup 4 > $O/stats_cfg4.log 2>&1
rm -f $O/stats_cfg4/*/r_kernel_trace.csv
cat $O/bench_cfg2.json $O/bench_cfg4.json
