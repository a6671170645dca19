#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c10
mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_free_run_gpu.py tests/test_mask3dformer.py tests/test_marching_cubes.py tests/test_scene_fusion_gpu.py -x -q -s > $O/new.log 2>&1; echo "new rc=$?" >> $O/new.log
grep -E "stage [0-9]|passed|failed|Error|assert" $O/new.log | head -30
timeout 300 python -m pytest tests/test_cfg4_gpu.py -x -q -k "exchange or pipelined" > $O/cfg4.log 2>&1; echo "cfg4 rc=$?" >> $O/cfg4.log
tail -4 $O/cfg4.log | cut -c1-250
for xs in 1 0; do
EPRECON_XCHG_STREAM=$xs EPRECON_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_dist_$xs.json 2> $O/bench_dist_$xs.err
python - $xs <<'PY'
import json,sys
b=json.loads([l for l in open(f'/root/repo/gpurun_out/r03_c10/bench_dist_{sys.argv[1]}.json') if l.startswith('{')][-1])
e=b.get('extra',{})
print('XCHG_STREAM='+sys.argv[1], {k:round(v,2) if isinstance(v,float) else v for k,v in e.items() if 'cfg5' in k and 'workload' not in k}, 'cfg4', round(e.get('cfg4_ms_per_fragment',0),2), 'cfg3', round(e.get('cfg3_ms_per_fragment',0),2), 'e2e', e.get('e2e_ms_per_fragment'), 'train', e.get('train_ms_per_step'), e.get('train_early_returns'))
PY
done
