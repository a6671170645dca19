#!/bin/bash
# Round-3 profile collection on the GPU box: bench line, kernel stats of the default bench command, PMC passes
# (each counter set in its own run, --kernel-trace only), cfg4 kernel stats -> gpurun_out/r03_final/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $O/stats.log 2>&1
rm -f $O/stats/r_kernel_trace.csv
for c in "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
  d=$O/pmc_$(echo $c | cut -c1-18 | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $d.log 2>&1
  rm -f $d/r_kernel_trace.csv
done
python $R/bench.py --workload cfg4 --steps 24 --warmup 8 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -o r -- python $R/bench.py --workload cfg4 --steps 12 --warmup 4 > $O/stats_cfg4.log 2>&1
cd $R
python tools/summarize_cfg4.py $O/stats_cfg4 $O/profiles_r03 $O/bench_cfg4.json > /dev/null
rm -f $O/stats_cfg4/r_kernel_trace.csv
python tools/summarize_profiles.py $O $O/profiles_r03 r03_final 35 | head -30
python tools/profile_cfg4_stages.py 3 > $O/profiles_r03/cfg4_stage_times.txt 2>&1
ls $O/profiles_r03
