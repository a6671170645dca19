#!/bin/bash
# cfg2 bench line + kernel stats of the same command (round 3)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $O/stats.log 2>&1
rm -f $O/stats/r_kernel_trace.csv
cd $R
python tools/summarize_profiles.py $O $O/profiles r03_mid 35 | head -45
EPRECON_CONV_DENSE3D=2 python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('DENSE3D=2 ms_per_step', b['ms_per_step'])"
EPRECON_CONV_DENSE3D=0 python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('DENSE3D=0 ms_per_step', b['ms_per_step'])"
python tools/profile_cfg2_stages.py 2>/dev/null | tail -15
