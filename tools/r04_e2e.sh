#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r04_e2e}
mkdir -p $O
cd $R
python tools/profile_e2e.py 3 > $O/e2e_stage_times.txt 2>&1
tail -8 $O/e2e_stage_times.txt
