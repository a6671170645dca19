#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r04_bench}
mkdir -p $O
cd $R
python bench.py --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('cfg2 ms', d['ms_per_step'], 'value', d['value'], 'roofline', d['roofline']['frac'], 'conv', d.get('roofline_conv',{}).get('frac'))
e=d.get('extra',{})
for k in sorted(e):
    if 'workload' not in k and 'source' not in k: print(' ', k, e[k] if not isinstance(e[k],dict) else {kk:e[k][kk] for kk in ('frac','avg_launch_ms','kernel') if kk in e[k]})
print('cpu', d.get('cpu_baseline',{}).get('value'))
"
tail -5 $O/bench.err
