#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c7
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_bn_fused_gpu.py tests/test_dense_conv3d_gpu.py -x -q > $O/new.log 2>&1; echo "new rc=$?" >> $O/new.log
tail -25 $O/new.log
timeout 700 python -m pytest tests -m gpu -q --deselect tests/test_bn_fused_gpu.py --deselect tests/test_dense_conv3d_gpu.py > $O/gpu.log 2>&1; echo "gpu rc=$?" >> $O/gpu.log
tail -12 $O/gpu.log | cut -c1-300
timeout 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
b=json.loads([l for l in open('/root/repo/gpurun_out/r03_c7/bench.json') if l.startswith('{')][-1])
print('ms_per_step', b['ms_per_step'], 'value', b['value'])
print('roofline_conv', b.get('roofline_conv'))
for k,v in b.get('extra',{}).items():
    if 'workload' not in k: print(k, v)
PY
tail -5 $O/bench.err
EPRECON_BN_TICKET=0 timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('BN_TICKET=0 ms_per_step', b['ms_per_step'])"
