#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "EPRECON_CFG4_PIPELINE=0" "EPRECON_CFG4_PIPELINE=0 EPRECON_CONV_DIRECT=0" "EPRECON_CFG4_PIPELINE=0 EPRECON_CONV_DIRECT=0 EPRECON_CONV_SPLITK_PIPE=0 EPRECON_CONV_SPLITK_NARROW=0 EPRECON_CONV_SPLITK_WAVES=4" "EPRECON_PIPELINE_THREAD=0" "EPRECON_PIPELINE_THREAD=0 EPRECON_CONV_DIRECT=0" "A=1" "A=2"; do
  env $v timeout 300 python bench.py --workload cfg4 --steps 16 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cfg4 $v', round(d['ms_per_step'],2))"
done
