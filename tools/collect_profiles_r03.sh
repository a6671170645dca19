#!/bin/bash
# Round-3 profile collection on the GPU box: bench line, kernel stats of the default bench command, PMC passes
# (each counter set in its own run, --kernel-trace only), cfg4 kernel stats, stage times, per-layer cfg4 trace, per-shape
# convolution timings -> gpurun_out/r03_final/ (copy profiles_r03/* into profiles/r03/)
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
python tools/profile_cfg2_stages.py > $O/profiles_r03/cfg2_stage_times.txt 2>&1
# per-stage / per-layer picture of a cfg4 fragment (stage markers + convolution log joined with the kernel trace)
L=$O/layers; rm -rf $L; mkdir -p $L
EPRECON_NO_GRAPH=1 EPRECON_CONV_LOG=$L/conv.log rocprofv3 --kernel-trace --output-format csv -d $L -o r -- python tools/trace_cfg4_layers.py $L > $L/run.log 2>&1
python tools/summarize_cfg4_layers.py $L > $O/profiles_r03/cfg4_layers.txt 2> $L/sum.err
rm -f $L/r_kernel_trace.csv
# the 3x3x3 shapes of that fragment one by one: this round's kernels against round 2's selection
python tools/conv_shapes_ab.py round3 > $L/shapes_r3.txt 2>&1
EPRECON_CONV_DIRECT=0 EPRECON_CONV_WIDEK=0 EPRECON_CONV_SPLITK_BDIRECT=0 EPRECON_CONV_SPLITK_PIPE=0 EPRECON_CONV_SPLITK_NARROW=0 EPRECON_CONV_SPLITK_WAVES=4 python tools/conv_shapes_ab.py round2 > $L/shapes_r2.txt 2>&1
paste -d'|' <(cut -c1-62 $L/shapes_r2.txt) <(cut -c52-62 $L/shapes_r3.txt) | grep -v amdgpu.ids > $O/profiles_r03/conv_shapes.txt
ls $O/profiles_r03
