#!/bin/bash
# same-box A/B of an environment switch on the cfg4 (unpipelined) and cfg2 benches: tools/r04_ab.sh <out> "<ENV=VAL ...>"
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r04_ab}
mkdir -p $O
cd $R
for rep in 1 2; do
  for variant in base alt; do
    if [ $variant = alt ]; then E="$2"; else E=""; fi
    env $E EPRECON_CFG4_PIPELINE=0 python bench.py --workload cfg4 --steps 24 --warmup 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg4 $variant $rep', round(d['ms_per_step'],3))" | tee -a $O/ab.txt
    env $E python bench.py --steps 40 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg2 $variant $rep', round(d['ms_per_step'],4))" | tee -a $O/ab.txt
  done
done
