#!/bin/bash
# Round-6 profile collection on the GPU box -> gpurun_out/r06_final/profiles_r06/ (copied to profiles/r06/ and committed).
# Every number DESIGN.md section 7i cites comes from a file this script writes, or from the A/B files the round's experiments
# left under profiles/r06/ (conv_forms_ab.txt, conv_direct_ablate.txt, conv_stage_depth_ab.txt, bn_acc_ab.txt,
# mfma_shapes_probe.txt, scenes_per_gpu.txt).  Counter passes run on their own (--kernel-trace + --pmc only).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_final
P=$O/profiles_r06
rm -rf $O; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
# 1. the driver's command + kernel statistics of the same command + PMC passes of the dominant kernel
python $R/bench.py --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $O/stats.log 2>&1
rm -f $O/stats/r_kernel_trace.csv
for c in "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  d=$O/pmc_$(echo $c | cut -c1-18 | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $d.log 2>&1
  rm -f $d/r_kernel_trace.csv
done
cd $R
python tools/summarize_profiles.py $O $P r06_final 35 | head -16
# 2. cfg4 (whole NeuConNet.forward, unpipelined = the drop-in contract): bench line, kernel statistics, launches / fragment
cd /tmp
EPRECON_CFG4_PIPELINE=0 python $R/bench.py --workload cfg4 --steps 32 --warmup 8 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
EPRECON_CFG4_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -o r -- python $R/bench.py --workload cfg4 --steps 12 --warmup 4 > $O/stats_cfg4.log 2>&1
cd $R
python tools/summarize_cfg4.py $O/stats_cfg4 $P $O/bench_cfg4.json > /dev/null
rm -f $O/stats_cfg4/r_kernel_trace.csv
# 2b. K independent scenes on the one GPU: aggregate fragments/s and per-scene ms
{
  echo "# python bench.py --workload cfg4 --scenes-per-gpu K --steps 16 --warmup 8: K scene processes on ONE MI355X, started together"
  echo "# K | fragments/s over all scenes | ms per fragment by scene | start skew ms"
  for k in 1 2 3 4 6 8; do
    python bench.py --workload cfg4 --scenes-per-gpu $k --steps 16 --warmup 8 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['scenes_per_gpu'], '|', round(d['value'],1), '|', d['ms_per_fragment_by_scene'], '|', d['start_skew_ms'])"
  done
} > $P/scenes_per_gpu.txt 2>&1
# 3. stage times (sync around every stage)
python tools/profile_cfg4_stages.py 3 > $P/cfg4_stage_times.txt 2>&1
python tools/profile_cfg2_stages.py > $P/cfg2_stage_times.txt 2>&1
# 4. the convolution that leads the cfg4 profile, alone: HIP events + the rocprofv3 rows of exactly those launches
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/conv_instance -o r -- python $R/tools/conv_cfg4_instance.py > $O/conv_instance.log 2>&1
cd $R
python - > $P/conv_cfg4_instance.txt <<PY
import csv, glob, re
log = [l.rstrip() for l in open("$O/conv_instance.log") if l.startswith(("rows ", "algorithmic "))]
print("# tools/conv_cfg4_instance.py under rocprofv3 --kernel-trace (MI355X)")
print("\n".join(log))
rows = int(re.search(r"rows (\d+)", log[0]).group(1))
trace = list(csv.DictReader(open(glob.glob("$O/conv_instance/**/*kernel_trace.csv", recursive=True)[0])))
trace.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = max(i for i, r in enumerate(trace) if "profile_mark" in r["Kernel_Name"] and int(r["Grid_Size_X"]) == 64 * 64)
sel = [r for r in trace[mark + 1:] if "spconv_direct16_kernel<2, 3, true" in r["Kernel_Name"] and int(r["Grid_Size_X"]) == (rows + 127) // 128 * 256]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in sel]
print(f"rocprofv3 kernel trace: {len(d)} launches of {sel[0]['Kernel_Name'][:66]} with grid {sel[0]['Grid_Size_X']}: avg {sum(d) / len(d):.1f} us, min {min(d):.1f}, max {max(d):.1f}")
PY
rm -rf $O/conv_instance/*/*kernel_trace.csv $O/conv_instance/*kernel_trace.csv
python tools/conv_shapes_ab.py round6 2>/dev/null > $P/conv_shapes.txt
EPRECON_CONV_TAIL8=0 EPRECON_CONV_STAGE_DEPTH=0 python tools/conv_shapes_ab.py "round6 build with the round-5 rules (padded tail, deepest stage)" 2>/dev/null > $P/conv_shapes_r05_rules.txt
EPRECON_CONV_BF16X3=1 python tools/conv_shapes_ab.py "round6, opt-in bf16x3 operand form of the direct gather kernel" 2>/dev/null > $P/conv_shapes_bf16x3.txt
bash tools/probes/interleave_ab.sh > $P/conv_interleave_ab.txt 2>/dev/null
bash tools/probes/splitk_fast_ab.sh > $P/conv_splitk_flat_ab.txt 2>/dev/null
python tools/conv_tail_ab.py --instance 2>/dev/null | grep -v calibration > $P/conv_forms_ab_final.txt
# 5. where the HOST time of a cfg4 fragment goes; how much of a fragment's wall time the GPU is busy (+ CU-level occupancy)
python tools/profile_cfg4_host.py 8 2>/dev/null > $P/cfg4_host_profile.txt
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/busy -o r -- python $R/tools/gpu_busy_cfg4.py > /dev/null 2>&1
cd $R
python tools/gpu_busy_cfg4.py --summarize $(find $O/busy -name "*kernel_trace.csv" | head -1) > $P/cfg4_gpu_busy.txt
rm -f $(find $O/busy -name "*kernel_trace.csv")
# 6. per-stage / per-layer kernel accounting (stage markers + convolution log joined with the kernel trace; graphs off)
bash tools/r04_layers.sh r06_layers > /dev/null 2>&1
cp $R/gpurun_out/r06_layers/cfg4_layers.txt $P/cfg4_layers.txt
# 7. the optimisation step: blocking reads by call site, host profile
python tools/profile_train_syncs.py 2>&1 | grep -v -e amdgpu.ids -e "prototype feature" -e _cuda_set_sync_debug_mode > $P/train_syncs.txt
python tools/profile_train_host.py 5 2>&1 | grep -v amdgpu.ids | head -45 > $P/train_host_profile.txt
# static: registers / LDS / scratch / waves per SIMD of every kernel, from the code objects' metadata (no GPU)
python tools/kernel_resources.py > $P/kernel_resources.txt
ls -la $P
