#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c9
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_exchange_gpu.py tests/test_gru_fusion_gpu.py tests/test_scene_fusion_gpu.py -x -q > $O/new.log 2>&1; echo "new rc=$?" >> $O/new.log
tail -15 $O/new.log | cut -c1-250
timeout 300 python -m pytest tests/test_cfg4_gpu.py -x -q > $O/cfg4.log 2>&1; echo "cfg4 rc=$?" >> $O/cfg4.log
tail -6 $O/cfg4.log | cut -c1-250
EPRECON_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "rc=$?"
python - <<'PY'
import json
b=json.loads([l for l in open('/root/repo/gpurun_out/r03_c9/bench_dist1.json') if l.startswith('{')][-1])
e=b.get('extra',{})
print({k:v for k,v in e.items() if 'cfg5' in k and 'workload' not in k}, 'cfg4', e.get('cfg4_ms_per_fragment'))
PY
tail -3 $O/bench_dist1.err
