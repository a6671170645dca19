#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c8
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_bn_fused_gpu.py tests/test_cfg4_gpu.py tests/test_spvcnn_gpu.py -x -q > $O/new.log 2>&1; echo "new rc=$?" >> $O/new.log
tail -12 $O/new.log | cut -c1-250
for t in 1 0; do
  for th in 1 0; do
    EPRECON_BN_TICKET=$t EPRECON_PIPELINE_THREAD=$th timeout 200 python bench.py --workload cfg4 --steps 24 --warmup 8 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('BN_TICKET=$t THREAD=$th cfg4 ms/fragment', round(b['ms_per_step'],2), b.get('finest_voxels_min_max'))" | tee -a $O/cfg4_ab.txt
  done
done
EPRECON_BN_TICKET=0 timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('cfg2 BN_TICKET=0 ms_per_step', b['ms_per_step'])" | tee -a $O/cfg4_ab.txt
EPRECON_BN_TICKET=1 timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('cfg2 BN_TICKET=1 (2D stack separate) ms_per_step', b['ms_per_step'])" | tee -a $O/cfg4_ab.txt
cd /tmp && export TMPDIR=/tmp
EPRECON_PIPELINE_THREAD=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -o r -- python $R/bench.py --workload cfg4 --steps 12 --warmup 4 > $O/stats_cfg4.log 2>&1
cd $R
EPRECON_PIPELINE_THREAD=0 timeout 100 python bench.py --workload cfg4 --steps 12 --warmup 4 > $O/bench_cfg4.json 2>/dev/null
python tools/summarize_cfg4.py $O/stats_cfg4 $O/profiles_r03 $O/bench_cfg4.json | head -5
rm -f $O/stats_cfg4/r_kernel_trace.csv
cat $O/profiles_r03/cfg4_kernel_stats.json
