#!/bin/bash
# full GPU test suite + default bench + cfg4 stage times -> gpurun_out/$1/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r04_full}
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --maxfail 15 --timeout 600 > $O/pytest.log 2>&1
tail -4 $O/pytest.log
bash tools/r04_bench.sh $1
timeout 600 python tools/profile_cfg4_stages.py 3 > $O/cfg4_stage_times.txt 2>&1
tail -23 $O/cfg4_stage_times.txt
