R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06h
rm -rf $O; mkdir -p $O/p
cd /tmp && export TMPDIR=/tmp
EPRECON_CFG4_PIPELINE=0 python $R/bench.py --workload cfg4 --steps 32 --warmup 8 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
EPRECON_CFG4_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -o r -- python $R/bench.py --workload cfg4 --steps 12 --warmup 4 > $O/stats_cfg4.log 2>&1
cd $R
python tools/summarize_cfg4.py $O/stats_cfg4 $O/p $O/bench_cfg4.json > /dev/null
rm -f $O/stats_cfg4/r_kernel_trace.csv
head -5 $O/p/bench_cfg4_rocprof_summary.txt
