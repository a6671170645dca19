#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')" 2>&1 | tail -1
for v in "0 0" "1 0" "1 1" "1 -1"; do
  set -- $v
  EPRECON_CFG2_BP_STREAM=$1 EPRECON_CFG2_BP_PRIO=$2 python bench.py --steps 40 --warmup 8 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('BP_STREAM=$1 PRIO=$2 ms_per_step', round(b['ms_per_step'],4), 'gather us', round(b['roofline']['avg_launch_ms']*1e3,1))"
done
