#!/bin/bash
# Round-5 profile collection on the GPU box -> gpurun_out/r05_final/profiles_r05/ (copied to profiles/r05/ and committed).
# Every number DESIGN.md section 7g cites comes from a file this script writes.  Counter passes run on their own
# (--kernel-trace + --pmc only).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_final
P=$O/profiles_r05
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
python tools/summarize_profiles.py $O $P r05_final 35 | head -16
# 1b. the gather's tile order: one contiguous slab of the raster per XCD (default) against tiles in hardware
#     block order — launch time by HIP events and HBM-side fetch (raw FETCH_SIZE, KiB) of the dense 96^3 launch
{
  echo "# bp_gather_mlp_kernel<256,MEAN,6,1> on the dense 96^3 level: EPRECON_BP_XCD_SLABS=1 (default: XCD k walks tiles [k n/8, (k+1) n/8), i.e. an x-slab of the raster) vs 0 (tile = hardware block id)"
  for v in 1 0; do
    EPRECON_BP_XCD_SLABS=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('slabs=$v  gather', round(d['roofline']['avg_launch_ms']*1e3,1), 'us  step', round(d['ms_per_step'],3), 'ms')"
    ( cd /tmp; EPRECON_BP_XCD_SLABS=$v rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_order$v -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/pmc_order$v.log 2>&1 )
    python - <<PY
import csv, glob
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(glob.glob("$O/pmc_order$v/**/*counter_collection.csv", recursive=True)[0]))
     if "bp_gather_mlp_kernel<256" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print(f"slabs=$v  FETCH_SIZE raw {sum(v) / len(v):.0f} KiB per launch (n={len(v)}) -> x2 gfx950 correction {2 * sum(v) / len(v) * 1024 / 1e6:.1f} MB")
PY
    rm -f $O/pmc_order$v/r_kernel_trace.csv $O/pmc_order$v/*/r_kernel_trace.csv
  done
} > $P/bp_tile_order_ab.txt 2>&1
# 2. cfg4 (whole NeuConNet.forward, unpipelined = the drop-in contract): bench line, kernel statistics, launches / fragment
cd /tmp
EPRECON_CFG4_PIPELINE=0 python $R/bench.py --workload cfg4 --steps 32 --warmup 8 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
EPRECON_CFG4_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -o r -- python $R/bench.py --workload cfg4 --steps 12 --warmup 4 > $O/stats_cfg4.log 2>&1
cd $R
python tools/summarize_cfg4.py $O/stats_cfg4 $P $O/bench_cfg4.json > /dev/null
rm -f $O/stats_cfg4/r_kernel_trace.csv
# the switches of this round: separate runs (drift between runs of ONE build is visible here), then interleaved in one process
for v in "default A=1" "round4_like EPRECON_SPVCNN_NATIVE=0 EPRECON_PREFETCH=0" "default_again A=1"; do
  set -- $v; name=$1; shift
  env "$@" EPRECON_CFG4_PIPELINE=0 python bench.py --workload cfg4 --steps 32 --warmup 8 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$name', '$*', round(d['ms_per_step'],3), 'ms/fragment', d['blocking_reads_per_fragment'], 'blocking reads')"
done > $P/cfg4_switches_separate_runs.txt 2>&1
python tools/ab_cfg4.py 12 > $P/cfg4_switches_ab.txt 2>/dev/null
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
sel = [r for r in trace[mark + 1:] if "spconv_direct16_kernel<2, 3, 3>" in r["Kernel_Name"] and int(r["Grid_Size_X"]) == (rows + 127) // 128 * 256]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in sel]
print(f"rocprofv3 kernel trace: {len(d)} launches of {sel[0]['Kernel_Name'][:60]} with grid {sel[0]['Grid_Size_X']}: avg {sum(d) / len(d):.1f} us, min {min(d):.1f}, max {max(d):.1f}")
PY
rm -rf $O/conv_instance/*/*kernel_trace.csv $O/conv_instance/*kernel_trace.csv
python tools/conv_shapes_ab.py round5 2>/dev/null > $P/conv_shapes.txt
EPRECON_AB_IN_AFFINE=1 python tools/conv_shapes_ab.py "round5, pending BatchNorm + ReLU on the input" 2>/dev/null > $P/conv_shapes_in_affine.txt
EPRECON_AB_IN_AFFINE=1 EPRECON_LIB_PATH=$R/eprecon_amd/libeprecon_hip_plain.so python tools/conv_shapes_ab.py "round5, pending BatchNorm + ReLU on the input, built WITHOUT -fno-honor-nans" 2>/dev/null > $P/conv_shapes_in_affine_plain.txt
# 5. the vector-L1 segment rate (what roofline.l1.peak cites) and the FETCH_SIZE calibration on the gather's access shape
tools/probes/l1_line_rate > $P/l1_line_rate.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $O/probe_pmc -o r -- $R/tools/probes/l1_line_rate > $O/probe_pmc.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib -o r -- $R/tools/probes/l1_line_rate calib > $O/calib.log 2>&1
cd $R
python - >> $P/l1_line_rate.txt <<PY
import csv, glob, re
# the probe's own rate in the UNIT the gather's l1 record is counted in (TCP_TOTAL_CACHE_ACCESSES of the PMC pass): launches of
# probe_kernel<0> in dispatch order are 7 x L1-resident, 7 x L2-resident, 7 x 16.6 MB; time per launch from the run without counters
rows = [r for r in csv.DictReader(open(glob.glob("$O/probe_pmc/**/*counter_collection.csv", recursive=True)[0])) if "probe_kernel<0>" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r.get("Dispatch_Id") or r.get("Start_Timestamp") or 0))
acc = [float(r["Counter_Value"]) for r in rows[:7]]
plain = open("$P/l1_line_rate.txt").read().split("## L1-resident")[1]
us = float(re.search(r"mode 0:.*?([0-9.]+) us", plain).group(1))
per_launch = sum(acc) / len(acc)
print(f"# rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum: probe_kernel<0>, L1-resident: {per_launch:.0f} accesses per launch "
      f"({per_launch / (2048 * 4 * 4096):.2f} per wave-level load) in {us:.1f} us (run without counters)")
print(f"# PMC-calibrated peak (mode 0, L1-resident): {per_launch / (us * 1e-6) / 1e9:.1f} G accesses/s = "
      f"{per_launch / (us * 1e-6) / (256 * 2.4e9):.3f} per clock per CU")
print(open("$O/calib.log").read().strip().splitlines()[-1])
for r in csv.DictReader(open(glob.glob("$O/calib/**/*counter_collection.csv", recursive=True)[0])):
    if "calib_" in r["Kernel_Name"]:
        print(f"# rocprofv3 --pmc FETCH_SIZE: {r['Kernel_Name'].split('(')[0]} | {r['Counter_Name']} | {float(r['Counter_Value']):.1f} (raw counter, KiB)")
PY
# 6. where the HOST time of a cfg4 fragment goes; how much of a fragment's wall time the GPU is busy
python tools/profile_cfg4_host.py 8 2>/dev/null > $P/cfg4_host_profile.txt
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/busy -o r -- python $R/tools/gpu_busy_cfg4.py > /dev/null 2>&1
cd $R
python tools/gpu_busy_cfg4.py --summarize $(find $O/busy -name "*kernel_trace.csv" | head -1) > $P/cfg4_gpu_busy.txt
rm -f $(find $O/busy -name "*kernel_trace.csv")
# 7. per-stage / per-layer kernel accounting (stage markers + convolution log joined with the kernel trace; graphs off)
bash tools/r04_layers.sh r05_layers > /dev/null 2>&1
cp $R/gpurun_out/r05_layers/cfg4_layers.txt $P/cfg4_layers.txt
# static: registers / LDS / scratch / waves per SIMD of every kernel, from the code objects' metadata (no GPU)
python tools/kernel_resources.py > $P/kernel_resources.txt
ls -la $P
